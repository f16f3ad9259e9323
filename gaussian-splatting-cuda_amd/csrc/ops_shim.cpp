// ops_shim.cpp — `namespace gsplat` on at::Tensor over the C ABI of libgsx.so.
//
// Mirrors the reference's host wrappers (gsplat/SphericalHarmonics.cpp, Intersect.cpp, Projection.cpp,
// Rasterization.cpp): same checks (device + contiguity, like CHECK_INPUT in gsplat/Common.h:12-17),
// same output allocation on the input's device, same error type (c10::Error via TORCH_CHECK), launches on
// the current HIP stream.  The only blocking point is, as upstream (Intersect.cpp:76), the read of
// n_isects inside intersect_tile — the op's return type needs the exact length.
// Also exports the ops to Python (pybind11 module `_gsx_ops`) for the tests and the bench.
#include <ATen/ATen.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#ifndef GSX_NO_PYBIND
#include <torch/extension.h>
#endif

#include <ATen/hip/HIPEvent.h>

#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <tuple>

#include "../../include/gsx.h"
#include "../../include/gsx_ops.h"
#include "../../include/gsx_training_ops.h"

namespace {

#define GSX_CHECK_INPUT(x)                                         \
    TORCH_CHECK((x).is_cuda(), #x " must be a CUDA tensor");       \
    TORCH_CHECK((x).is_contiguous(), #x " must be contiguous")

// the reference's DEVICE_GUARD (Common.h:18-19); the is_cuda check comes first so that a CPU tensor gets CHECK_INPUT's message
// instead of the guard's internal assert
#define GSX_DEVICE_GUARD(x)                                    \
    TORCH_CHECK((x).is_cuda(), #x " must be a CUDA tensor");   \
    const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(at::device_of(x))

// the shim was compiled against include/gsx.h: refuse to run on a libgsx.so with another ABI (a stale library silently shifts arguments)
const bool abi_checked = [] {
    TORCH_CHECK(gsx_abi_version() == GSX_ABI_VERSION, "libgsx.so has ABI version ", gsx_abi_version(), ", this shim was built against ", GSX_ABI_VERSION,
                " (include/gsx.h): rebuild both");
    return true;
}();

inline void* cur_stream() { return (void*)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream(); }
inline void check(int rc, const char* op) { TORCH_CHECK(rc == GSX_OK, op, " failed (", rc, "): ", gsx_last_error()); }

inline const float* fptr(const at::optional<at::Tensor>& t) {
    return (t.has_value() && t->defined() && t->numel() > 0) ? t->data_ptr<float>() : nullptr;
}
inline const uint8_t* bptr(const at::optional<at::Tensor>& t) {
    return (t.has_value() && t->defined() && t->numel() > 0) ? reinterpret_cast<const uint8_t*>(t->data_ptr<bool>()) : nullptr;
}

gsx_cameras make_cams(const at::Tensor& viewmats0, const at::optional<at::Tensor>& viewmats1, const at::Tensor& Ks,
                      gsplat::CameraModelType model, ShutterType rs, const at::optional<at::Tensor>& radial,
                      const at::optional<at::Tensor>& tangential, const at::optional<at::Tensor>& thin_prism, uint32_t C) {
    GSX_CHECK_INPUT(viewmats0);
    GSX_CHECK_INPUT(Ks);
    TORCH_CHECK(viewmats0.scalar_type() == at::kFloat && Ks.scalar_type() == at::kFloat, "camera tensors must be float32");
    if (viewmats1.has_value()) { GSX_CHECK_INPUT(viewmats1.value()); }
    if (radial.has_value()) { GSX_CHECK_INPUT(radial.value()); }
    if (tangential.has_value()) { GSX_CHECK_INPUT(tangential.value()); }
    if (thin_prism.has_value()) { GSX_CHECK_INPUT(thin_prism.value()); }
    gsx_cameras c;
    c.C = C;
    c.viewmats0 = viewmats0.data_ptr<float>();
    c.viewmats1 = fptr(viewmats1);
    c.Ks = Ks.data_ptr<float>();
    c.camera_model = (int32_t)model;
    c.shutter = (int32_t)rs;
    c.radial = fptr(radial);
    // The pinhole model reads SIX radial coefficients per camera (Cameras.cuh: OpenCVPinholeCameraModel, k1 .. k6 of the rational model); the reference's glue pads what a
    // camera carries to FOUR (rasterizer.cpp:183-194: a COLMAP OPENCV camera has k1, k2) and its kernels then read two floats past the tensor.  Here the missing
    // coefficients are zeros: the tensor is padded to six per camera (kept alive until this thread's next call: the launch that follows reads it).
    if (model == gsplat::PINHOLE && radial.has_value() && radial->defined() && radial->numel() > 0 && radial->numel() < (int64_t)6 * C && radial->numel() % C == 0) {
        static thread_local at::Tensor padded;
        const int64_t per = radial->numel() / C;
        padded = at::constant_pad_nd(radial->reshape({(int64_t)C, per}), {0, 6 - per}, 0).contiguous();
        c.radial = padded.data_ptr<float>();
    }
    c.tangential = fptr(tangential);
    c.thin_prism = fptr(thin_prism);
    return c;
}
gsx_ut_params make_ut(const UnscentedTransformParameters& u) {
    return gsx_ut_params{u.alpha, u.beta, u.kappa, u.in_image_margin_factor, u.require_all_sigma_points_valid ? 1 : 0};
}

}  // namespace

// Side channel of the fused render path (gsx_ext::rasterize_fwd_keep_ws / bwd with fwd_ws): the blend forward hands its
// workspace (packed per-Gaussian records) to the caller, the backward of the same inputs takes it back and skips re-packing.
static thread_local at::Tensor* g_fwd_ws_out = nullptr;
static thread_local const at::Tensor* g_fwd_ws_in = nullptr;
// rasterize_bwd_act (fused render path): the raw SplatData tensors and the raw-gradient outputs of the activation epilogue (include/gsx.h, ABI 7)
struct ActArgs { at::Tensor scaling_raw, rotation_raw, opacity_raw, v_scaling_raw, v_rotation_raw, v_opacity_raw; float scale_reg, opacity_reg; };
static thread_local const ActArgs* g_act = nullptr;
// ... and the fused front end (gsx_ext::frontend_fused) hands the blend forward a workspace whose records are already packed
static thread_local const at::Tensor* g_fwd_ws_ready = nullptr;

// Round 6: two per-thread caches behind the PLAIN Ops.h entry points — what a reference build that swapped in this backend calls, one operator at a
// time (the reference's own render call site on the drop-in: 83 torch launches + 14 gsx launches per frame, profiles/r06_dropin_callers.md).
// They only save work the operator sequence of gs::training::rasterize repeats; results are the same bits.  A cache holds REFERENCES to the
// tensors it is keyed by (their storage cannot be freed and handed to another tensor meanwhile) and compares data pointers, sizes and autograd
// version counters; GSX_SHIM_CACHE=0 (test switch) turns both off.
static bool shim_cache_on() {
    static const bool on = [] { const char* e = gsx_test_switch("GSX_SHIM_CACHE"); return !(e && e[0] == '0'); }();
    return on;
}
struct TensorKey {
    at::Tensor t; const void* ptr = nullptr; uint32_t version = 0; int64_t numel = 0;
    static uint32_t ver(const at::Tensor& x) { return x.is_inference() ? 0u : (uint32_t)x._version(); }   // (inference tensors keep no version counter)
    void set(const at::Tensor& x) { t = x; ptr = x.defined() ? x.data_ptr() : nullptr; version = x.defined() ? ver(x) : 0; numel = x.defined() ? x.numel() : 0; }
    bool matches(const at::Tensor& x) const { return t.defined() && x.defined() && x.data_ptr() == ptr && x.numel() == numel && ver(x) == version; }
    void reset() { t = at::Tensor(); ptr = nullptr; }
};
// (a) intersect_tile through the binned pipeline has isect_offsets as a by-product; the reference asks for them in a second call,
//     intersect_offset(isect_ids, ...) (rasterizer.cpp:305-329): answered from here when it comes with the very isect_ids tensor (one
//     lower_bound launch over 27 MB of keys less per frame)
struct OffsetsCache { TensorKey ids; at::Tensor offsets; uint32_t C = 0, tw = 0, th = 0; };
static thread_local OffsetsCache g_offsets_cache;
// (b) the blend forward packs one 64 B record per (camera, Gaussian) into its workspace; the backward of the same inputs
//     (rasterizer_autograd.cpp:331-391 hands back the tensors the forward saved) takes the records from here instead of packing them again
struct PackCache {
    TensorKey means, quats, scales, colors, opacities, viewmats0, Ks;
    at::Tensor fws; uint32_t W = 0, H = 0, C = 0, N = 0; int camera_model = -1, shutter = -1; bool distorted = false;
    void reset() { means.reset(); fws = at::Tensor(); }
};
// (process-wide, under a mutex: autograd runs a CUDA node's backward on its device thread, not on the thread that ran the forward)
static PackCache g_pack_cache;
static std::mutex g_pack_cache_mutex;

// counters for bench.py (host synchronisations and capacity-hint outcomes of intersect_tile); never read by the ops themselves
struct ShimStats { std::atomic<int64_t> host_syncs{0}, binned_calls{0}, hint_misses{0}, hint_cold{0}, ranked_calls{0}, guarded_calls{0}, guarded_waits{0}, guarded_misses{0}; };
static ShimStats g_stats;


// ---- capacity hints of the binned intersection -------------------------------------------------------------------------------
// The ONLY state the shim keeps between calls: a process-wide map {(device, C, N, tile grid) -> recent maxima of n_isects and of the
// largest tile segment}, mutex protected, over a sliding window of frames so that one outlier view does not pin memory for ever.  It never changes a
// result: it sizes the optimistic fill (exact protocol: outputs narrowed / fill repeated on a miss; guarded protocol: the frame is
// rendered again on a miss).  A model that grows (densification changes N every few hundred iterations) would start cold after every
// resize: the entry with N = 0 holds the last call of the same (device, C, tile grid) at any N and stands in, scaled by the ratio of
// the Gaussian counts.
using HintKey = std::tuple<int, uint32_t, uint32_t, uint32_t, uint32_t>;
static std::mutex g_hint_mutex;
static std::map<HintKey, std::pair<int64_t, int64_t>> g_hints;   // (n_isects, largest segment)
static std::map<HintKey, std::pair<int64_t, int64_t>> g_last;    // the last confirmed frame of the shape, undecayed (launch decisions)
static std::map<HintKey, uint32_t> g_hint_last_n;
struct HintWindow { int64_t cur_n, cur_seg, prev_n, prev_seg; uint32_t count; };
static std::map<HintKey, HintWindow> g_hint_win;                 // the two windows behind g_hints' maxima (hint_update)

static HintKey hint_key_any(const HintKey& k) { return std::make_tuple(std::get<0>(k), std::get<1>(k), 0u, std::get<3>(k), std::get<4>(k)); }

// (g_hint_mutex held) the hint of `key`: its own running maxima, else those of the same (device, C, tile grid) at the previous Gaussian count
static void hint_lookup_locked(const HintKey& key, int64_t& hint, int64_t& hint_seg) {
    const HintKey any = hint_key_any(key);
    auto it = g_hints.find(key);
    hint = hint_seg = 0;
    if (it != g_hints.end()) {
        hint = it->second.first; hint_seg = it->second.second;
    } else if ((it = g_hints.find(any)) != g_hints.end() && g_hint_last_n[any] > 0) {
        // the last call of this (device, C, tile grid) at another Gaussian count stands in, scaled — if the counts are close (a growth step);
        // a model 1.5x larger or smaller than that call's is a different scene: no hint, the exact protocol runs once
        const double ratio = (double)std::get<2>(key) / (double)g_hint_last_n[any];
        if (ratio <= 1.5 && ratio >= 0.67) {
            const double grow = std::max(1.0, ratio);
            // ... and what stands in is that shape's running MAXIMUM over its cameras (its last call alone may have been a light view)
            auto prev = g_hints.find(std::make_tuple(std::get<0>(key), std::get<1>(key), g_hint_last_n[any], std::get<3>(key), std::get<4>(key)));
            const std::pair<int64_t, int64_t> base = prev != g_hints.end() ? std::make_pair(std::max(prev->second.first, it->second.first), std::max(prev->second.second, it->second.second))
                                                                           : it->second;
            hint = (int64_t)((double)base.first * grow); hint_seg = (int64_t)((double)base.second * grow);
        }
    }
}

static void hint_lookup(const HintKey& key, int64_t& hint, int64_t& hint_seg, int64_t* last_total = nullptr) {
    std::lock_guard<std::mutex> lock(g_hint_mutex);
    hint_lookup_locked(key, hint, hint_seg);
    if (last_total) {
        auto lt = g_last.find(key);
        *last_total = lt != g_last.end() ? lt->second.first : hint;
    }
}

static void hint_update(const HintKey& key, int64_t n_isects, int64_t max_seg) {
    std::lock_guard<std::mutex> lock(g_hint_mutex);
    if (g_hints.size() > 4096) { g_hints.clear(); g_last.clear(); g_hint_win.clear(); }   // (a long run that resizes thousands of times: start over rather than grow without bound)
    if (g_hints.find(key) == g_hints.end()) {   // a new shape inherits what stood in for it (a grown model keeps the maxima over its cameras)
        int64_t h0 = 0, s0 = 0;
        hint_lookup_locked(key, h0, s0);
        g_hints[key] = std::make_pair(h0, s0);
    }
    auto& h = g_hints[key];
    // Maxima over a sliding window of frames: the hint is the maximum of the current and the previous window of kHintWindow confirmed frames,
    // so a heavy view is remembered for 1024 - 2048 frames — at least one round of a dataset's cameras — and an outlier is forgotten after that.
    // (Rounds 2 - 4 decayed the maximum by 1/512 per frame: a view 1.5x the median was forgotten within ~200 frames, i.e. missed once per
    // epoch on a dataset of a few hundred views; under the guarded protocol a miss costs a whole repeated iteration.  ADVICE r04.)
    constexpr uint32_t kHintWindow = 1024;
    auto wi = g_hint_win.find(key);
    if (wi == g_hint_win.end()) wi = g_hint_win.emplace(key, HintWindow{0, 0, h.first, h.second, 0}).first;   // what the shape inherited counts as the previous window
    HintWindow& w = wi->second;
    w.cur_n = std::max(w.cur_n, n_isects); w.cur_seg = std::max(w.cur_seg, max_seg);
    if (++w.count >= kHintWindow) { w.prev_n = w.cur_n; w.prev_seg = w.cur_seg; w.cur_n = n_isects; w.cur_seg = max_seg; w.count = 0; }
    h.first = std::max(w.cur_n, w.prev_n);
    h.second = std::max(w.cur_seg, w.prev_seg);
    const HintKey any = hint_key_any(key);
    g_hints[any] = std::make_pair(n_isects, max_seg);
    g_last[key] = std::make_pair(n_isects, max_seg);
    g_hint_last_n[any] = std::get<2>(key);
}

// Handle of one guarded intersection (include/gsx.h "guarded lists"): the lists were filled optimistically into `capacity` slots and the
// blend kernels read the verdict from `status` on the device; the host reads the same verdict here, whenever it likes, and must do so
// (confirm) before it applies anything irreversible to a frame rendered from these lists.  confirm() waits for the 8-byte word the count
// copied to pinned memory — by then hundreds of microseconds of queued kernels sit behind that copy, so the wait does not drain the stream —
// and feeds the capacity hints.
// The pinned host word of one count (n_isects | largest segment << 32): a slot of a process-wide ring of HOST-COHERENT pinned memory
// (hipHostMallocCoherent: the device's system-scope store is visible to the CPU without a release at kernel end).  The count leaves all
// ones in it until the device has written it, so the host POLLS the word: no event is recorded behind the count — an event's
// system-scope release between bin_scan and the key scatter cost ~6 us of idle GPU per frame (kernel trace, round 4).
struct HostWord {
    static constexpr uint64_t kSlots = 4096;
    volatile uint64_t* p = nullptr;
    uint64_t ticket = 0;
    static uint64_t* ring() {
        static uint64_t* base = [] {
            void* q = nullptr;
            TORCH_CHECK(hipHostMalloc(&q, kSlots * sizeof(uint64_t), hipHostMallocCoherent | hipHostMallocPortable | hipHostMallocMapped) == hipSuccess && q != nullptr,
                        "gsx: hipHostMalloc of the host-word ring failed");
            return (uint64_t*)q;
        }();
        return base;
    }
    static std::atomic<uint64_t>& counter() { static std::atomic<uint64_t> c{0}; return c; }
    static HostWord take() {
        HostWord w;
        w.ticket = counter().fetch_add(1);
        w.p = ring() + (w.ticket % kSlots);
        *w.p = ~0ull;
        return w;
    }
    bool ready() const { return (uint32_t)(__atomic_load_n(p, __ATOMIC_ACQUIRE) >> 32) != 0xFFFFFFFFu; }
    // waits for the device's store (by polling; a word that does not arrive within 20 s means the stream died: synchronise to surface the error)
    uint64_t wait() const {
        TORCH_CHECK(counter().load() - ticket <= kSlots, "gsx: an intersection handle was confirmed after ", kSlots, " later intersections (its host word was reused)");
        const auto t0 = std::chrono::steady_clock::now();
        uint32_t spins = 0;
        while (!ready()) {
            if ((++spins & 1023u) == 0u && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(20)) {
                C10_HIP_CHECK(hipDeviceSynchronize());
                TORCH_CHECK(ready(), "gsx: the intersection count never reached the host");
            }
#if defined(__x86_64__)
            __builtin_ia32_pause();
#endif
            if (spins > 8192u && (spins & 255u) == 0u) std::this_thread::yield();   // a long wait (cold start, a deep queue): leave the core to others now and then
        }
        return __atomic_load_n(p, __ATOMIC_ACQUIRE);
    }
};

struct IsectLists {
    HostWord n_host;      // pinned: n_isects | largest segment << 32
    at::Tensor status;    // device int32 [1]: n_isects, or -1 = lists incomplete (frame renders empty); undefined = exact lists (cold call)
    HintKey key;
    int64_t capacity = 0, seg_bound = 0, expected = 0;
    bool ranked = false, confirmed = false, complete = true;
    int64_t n_isects = 0, max_seg = 0;

    bool is_ready() const { return confirmed || n_host.ready(); }
    std::tuple<int64_t, int64_t, bool> confirm() {
        if (!confirmed) {
            if (!n_host.ready()) g_stats.guarded_waits++;   // the host got here before the GPU passed the count
            const uint64_t word = n_host.wait();
            n_isects = (int64_t)(word & 0xFFFFFFFFull); max_seg = (int64_t)(word >> 32);
            TORCH_CHECK(n_isects <= 0x7FFFFFFFll, "intersect_tile: more than 2^31 - 1 intersections (tile offsets are int32, as upstream's isect_offsets)");
            hint_update(key, n_isects, max_seg);
            complete = n_isects <= capacity && (ranked || max_seg <= seg_bound);
            if (!complete) { g_stats.hint_misses++; g_stats.guarded_misses++; }
            confirmed = true;
        }
        return std::make_tuple(n_isects, max_seg, complete);
    }
};
// side channel of the fused render path: the blend ops called next take their lists guarded by this handle (set by the Python bindings)
static thread_local const IsectLists* g_lists = nullptr;

namespace gsx_ext {
std::tuple<at::Tensor, at::Tensor, at::Tensor, at::Tensor> intersect_tile_binned(const at::Tensor means2d, const at::Tensor radii,
                                                                                 const at::Tensor depths, const uint32_t C,
                                                                                 const uint32_t tile_size, const uint32_t tile_width,
                                                                                 const uint32_t tile_height, const bool want_isect_ids);
}

namespace gsplat {

at::Tensor spherical_harmonics_fwd(const uint32_t degrees_to_use, const at::Tensor dirs, const at::Tensor coeffs,
                                   const at::optional<at::Tensor> masks) {
    GSX_DEVICE_GUARD(dirs);
    GSX_CHECK_INPUT(dirs);
    GSX_CHECK_INPUT(coeffs);
    if (masks.has_value()) { GSX_CHECK_INPUT(masks.value()); }
    TORCH_CHECK(coeffs.size(-1) == 3, "coeffs must have last dimension 3");
    TORCH_CHECK(dirs.size(-1) == 3, "dirs must have last dimension 3");
    TORCH_CHECK(dirs.scalar_type() == at::kFloat && coeffs.scalar_type() == at::kFloat, "float32 only");
    at::Tensor colors = at::empty_like(dirs);
    const uint32_t K = coeffs.size(-2), N = dirs.numel() / 3;
    check(gsx_spherical_harmonics_fwd(degrees_to_use, N, K, dirs.data_ptr<float>(), coeffs.data_ptr<float>(), bptr(masks),
                                      colors.data_ptr<float>(), cur_stream()),
          "spherical_harmonics_fwd");
    return colors;
}

std::tuple<at::Tensor, at::Tensor> spherical_harmonics_bwd(const uint32_t K, const uint32_t degrees_to_use,
                                                           const at::Tensor dirs, const at::Tensor coeffs,
                                                           const at::optional<at::Tensor> masks,
                                                           const at::Tensor v_colors, bool compute_v_dirs) {
    GSX_DEVICE_GUARD(dirs);
    GSX_CHECK_INPUT(dirs);
    GSX_CHECK_INPUT(coeffs);
    GSX_CHECK_INPUT(v_colors);
    if (masks.has_value()) { GSX_CHECK_INPUT(masks.value()); }
    TORCH_CHECK(v_colors.size(-1) == 3, "v_colors must have last dimension 3");
    TORCH_CHECK(coeffs.size(-1) == 3, "coeffs must have last dimension 3");
    TORCH_CHECK(dirs.size(-1) == 3, "dirs must have last dimension 3");
    TORCH_CHECK((int64_t)K == coeffs.size(-2), "K must equal coeffs.size(-2)");
    const uint32_t N = dirs.numel() / 3;
    at::Tensor v_coeffs = at::empty_like(coeffs);  // fully written by the kernel (incl. zeros)
    at::Tensor v_dirs;
    if (compute_v_dirs) v_dirs = at::empty_like(dirs);
    check(gsx_spherical_harmonics_bwd(K, degrees_to_use, N, dirs.data_ptr<float>(), coeffs.data_ptr<float>(), bptr(masks),
                                      v_colors.data_ptr<float>(), v_coeffs.data_ptr<float>(),
                                      compute_v_dirs ? v_dirs.data_ptr<float>() : nullptr, cur_stream()),
          "spherical_harmonics_bwd");
    return std::make_tuple(v_coeffs, v_dirs);
}

static std::tuple<at::Tensor, at::Tensor, at::Tensor> intersect_tile_impl(const at::Tensor means2d, const at::Tensor radii, const at::Tensor depths,
                                                                          const at::optional<at::Tensor> camera_ids,
                                                                          const at::optional<at::Tensor> gaussian_ids, const uint32_t C,
                                                                          const uint32_t tile_size, const uint32_t tile_width,
                                                                          const uint32_t tile_height, const bool sort, const bool allow_binned);

std::tuple<at::Tensor, at::Tensor, at::Tensor> intersect_tile(const at::Tensor means2d, const at::Tensor radii,
                                                              const at::Tensor depths,
                                                              const at::optional<at::Tensor> camera_ids,
                                                              const at::optional<at::Tensor> gaussian_ids,
                                                              const uint32_t C, const uint32_t tile_size,
                                                              const uint32_t tile_width, const uint32_t tile_height,
                                                              const bool sort) {
    return intersect_tile_impl(means2d, radii, depths, camera_ids, gaussian_ids, C, tile_size, tile_width, tile_height, sort, true);
}

static std::tuple<at::Tensor, at::Tensor, at::Tensor> intersect_tile_impl(const at::Tensor means2d, const at::Tensor radii, const at::Tensor depths,
                                                                          const at::optional<at::Tensor> camera_ids,
                                                                          const at::optional<at::Tensor> gaussian_ids, const uint32_t C,
                                                                          const uint32_t tile_size, const uint32_t tile_width,
                                                                          const uint32_t tile_height, const bool sort, const bool allow_binned) {
    GSX_DEVICE_GUARD(means2d);
    GSX_CHECK_INPUT(means2d);
    GSX_CHECK_INPUT(radii);
    GSX_CHECK_INPUT(depths);
    const bool packed = means2d.dim() == 2;   // [nnz, 2]: the packed layout (Intersect.cpp:31-38); the world-space blend itself is non-packed only
    if (packed) {
        TORCH_CHECK(camera_ids.has_value() && gaussian_ids.has_value(), "When packed is set, camera_ids and gaussian_ids must be provided.");
        GSX_CHECK_INPUT(camera_ids.value());
        GSX_CHECK_INPUT(gaussian_ids.value());
        TORCH_CHECK(camera_ids->scalar_type() == at::kLong && camera_ids->numel() == means2d.size(0), "camera_ids must be int64 [nnz]");
    }
    TORCH_CHECK(means2d.scalar_type() == at::kFloat && depths.scalar_type() == at::kFloat, "float32 only");
    TORCH_CHECK(radii.scalar_type() == at::kInt, "radii must be int32");
    const uint32_t n_elements = means2d.numel() / 2;
    const uint32_t N = packed ? n_elements : (C ? n_elements / C : 0);
    const uint32_t C_count = packed ? 1u : C;   // the count pass does not look at the camera: nnz pairs are counted as one camera's
    static const bool force_device_sort = [] { const char* e = gsx_test_switch("GSX_INTERSECT"); return e && std::string(e) == "sort"; }();
    if (!packed && allow_binned && sort && !force_device_sort && n_elements && gsx_intersect_bin_supported(tile_width, tile_height)) {
        // same three outputs through the binned pipeline (LDS histograms + per-tile LDS sort), ~2x faster than the device-wide sort
        auto r = gsx_ext::intersect_tile_binned(means2d, radii, depths, C, tile_size, tile_width, tile_height, true);
        if (shim_cache_on()) {   // the pipeline's isect_offsets, for the intersect_offset call that follows with these isect_ids
            g_offsets_cache.ids.set(std::get<1>(r));
            g_offsets_cache.offsets = std::get<3>(r);
            g_offsets_cache.C = C; g_offsets_cache.tw = tile_width; g_offsets_cache.th = tile_height;
        }
        return std::make_tuple(std::get<0>(r), std::get<1>(r), std::get<2>(r));
    }
    void* st = cur_stream();
    at::Tensor tiles_per_gauss = at::empty_like(depths, depths.options().dtype(at::kInt));
    int64_t n_isects = 0;
    at::Tensor cum;
    if (n_elements) {
        cum = at::empty({(int64_t)n_elements}, depths.options().dtype(at::kLong));
        const size_t wsb = gsx_intersect_count_workspace_bytes(C_count, N);
        at::Tensor ws = at::empty({(int64_t)wsb}, depths.options().dtype(at::kByte));
        at::Tensor n_host = at::empty({1}, at::TensorOptions().dtype(at::kLong).pinned_memory(true));
        check(gsx_intersect_tile_count(C_count, N, means2d.data_ptr<float>(), radii.data_ptr<int32_t>(), tile_size, tile_width,
                                       tile_height, tiles_per_gauss.data_ptr<int32_t>(), cum.data_ptr<int64_t>(), nullptr,
                                       n_host.data_ptr<int64_t>(), ws.data_ptr(), wsb, st),
              "intersect_tile(count)");
        c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().synchronize();  // the one host sync, as upstream (Intersect.cpp:76)
        g_stats.host_syncs++;
        n_isects = n_host.data_ptr<int64_t>()[0];
    }
    at::Tensor isect_ids = at::empty({n_isects}, depths.options().dtype(at::kLong));
    at::Tensor flatten_ids = at::empty({n_isects}, depths.options().dtype(at::kInt));
    if (n_isects) {
        const size_t wsb = gsx_intersect_fill_workspace_bytes(C, N, n_isects, sort ? 1 : 0);
        at::Tensor ws = at::empty({(int64_t)wsb}, depths.options().dtype(at::kByte));
        check(gsx_intersect_tile_fill_packed(C, N, packed ? n_elements : 0u, packed ? camera_ids->data_ptr<int64_t>() : nullptr,
                                             means2d.data_ptr<float>(), radii.data_ptr<int32_t>(), depths.data_ptr<float>(),
                                             cum.data_ptr<int64_t>(), tile_size, tile_width, tile_height, sort ? 1 : 0, n_isects,
                                             isect_ids.data_ptr<int64_t>(), flatten_ids.data_ptr<int32_t>(), ws.data_ptr(), wsb, st),
              "intersect_tile(fill)");
    }
    return std::make_tuple(tiles_per_gauss, isect_ids, flatten_ids);
}

// the reference's algorithm (one device-wide stable radix sort of all keys), kept for tile grids beyond the LDS counters and as
// the cross-check of the binned pipeline in the tests
std::tuple<at::Tensor, at::Tensor, at::Tensor> intersect_tile_device_sort(const at::Tensor means2d, const at::Tensor radii, const at::Tensor depths,
                                                                          const uint32_t C, const uint32_t tile_size, const uint32_t tile_width,
                                                                          const uint32_t tile_height, const bool sort) {
    return intersect_tile_impl(means2d, radii, depths, at::nullopt, at::nullopt, C, tile_size, tile_width, tile_height, sort, false);
}

at::Tensor intersect_offset(const at::Tensor isect_ids, const uint32_t C, const uint32_t tile_width,
                            const uint32_t tile_height) {
    GSX_DEVICE_GUARD(isect_ids);
    GSX_CHECK_INPUT(isect_ids);
    TORCH_CHECK(isect_ids.scalar_type() == at::kLong, "isect_ids must be int64");
    if (g_offsets_cache.ids.matches(isect_ids) && g_offsets_cache.C == C && g_offsets_cache.tw == tile_width && g_offsets_cache.th == tile_height &&
        isect_ids.numel() > 0) {
        at::Tensor cached = std::move(g_offsets_cache.offsets);   // handed out once: the caller owns it (a caller that writes into it cannot spoil a later answer)
        g_offsets_cache.ids.reset();
        g_offsets_cache.offsets = at::Tensor();
        if (cached.defined()) return cached;
    }
    at::Tensor offsets = at::empty({(int64_t)C, (int64_t)tile_height, (int64_t)tile_width}, isect_ids.options().dtype(at::kInt));
    check(gsx_intersect_offset(isect_ids.size(0), isect_ids.numel() ? isect_ids.data_ptr<int64_t>() : nullptr, C, tile_width,
                               tile_height, offsets.data_ptr<int32_t>(), cur_stream()),
          "intersect_offset");
    return offsets;
}

std::tuple<at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor> projection_ut_3dgs_fused(
    const at::Tensor means, const at::Tensor quats, const at::Tensor scales, const at::optional<at::Tensor> opacities,
    const at::Tensor viewmats0, const at::optional<at::Tensor> viewmats1, const at::Tensor Ks,
    const uint32_t image_width, const uint32_t image_height, const float eps2d, const float near_plane,
    const float far_plane, const float radius_clip, const bool calc_compensations, const CameraModelType camera_model,
    const UnscentedTransformParameters ut_params, ShutterType rs_type, const at::optional<at::Tensor> radial_coeffs,
    const at::optional<at::Tensor> tangential_coeffs, const at::optional<at::Tensor> thin_prism_coeffs) {
    GSX_DEVICE_GUARD(means);
    GSX_CHECK_INPUT(means);
    GSX_CHECK_INPUT(quats);
    GSX_CHECK_INPUT(scales);
    if (opacities.has_value()) { GSX_CHECK_INPUT(opacities.value()); }
    TORCH_CHECK(means.scalar_type() == at::kFloat, "float32 only");
    const uint32_t N = means.size(0), C = Ks.size(0);
    const gsx_cameras cams = make_cams(viewmats0, viewmats1, Ks, camera_model, rs_type, radial_coeffs, tangential_coeffs, thin_prism_coeffs, C);
    const gsx_ut_params ut = make_ut(ut_params);
    at::Tensor radii = at::empty({C, N, 2}, means.options().dtype(at::kInt));
    at::Tensor means2d = at::empty({C, N, 2}, means.options());
    at::Tensor depths = at::empty({C, N}, means.options());
    at::Tensor conics = at::empty({C, N, 3}, means.options());
    at::Tensor compensations;
    if (calc_compensations) compensations = at::zeros({C, N}, means.options());
    check(gsx_projection_ut_3dgs_fused(N, means.data_ptr<float>(), quats.data_ptr<float>(), scales.data_ptr<float>(),
                                       fptr(opacities), &cams, image_width, image_height, eps2d, near_plane, far_plane,
                                       radius_clip, &ut, radii.data_ptr<int32_t>(), means2d.data_ptr<float>(),
                                       depths.data_ptr<float>(), conics.data_ptr<float>(),
                                       calc_compensations ? compensations.data_ptr<float>() : nullptr, cur_stream()),
          "projection_ut_3dgs_fused");
    return std::make_tuple(radii, means2d, depths, conics, compensations);
}

std::tuple<at::Tensor, at::Tensor, at::Tensor> rasterize_to_pixels_from_world_3dgs_fwd(
    const at::Tensor means, const at::Tensor quats, const at::Tensor scales, const at::Tensor colors,
    const at::Tensor opacities, const at::optional<at::Tensor> backgrounds, const at::optional<at::Tensor> masks,
    const uint32_t image_width, const uint32_t image_height, const uint32_t tile_size, const at::Tensor viewmats0,
    const at::optional<at::Tensor> viewmats1, const at::Tensor Ks, const CameraModelType camera_model,
    const UnscentedTransformParameters ut_params, ShutterType rs_type, const at::optional<at::Tensor> radial_coeffs,
    const at::optional<at::Tensor> tangential_coeffs, const at::optional<at::Tensor> thin_prism_coeffs,
    const at::Tensor tile_offsets, const at::Tensor flatten_ids) {
    GSX_DEVICE_GUARD(means);
    GSX_CHECK_INPUT(means);
    GSX_CHECK_INPUT(quats);
    GSX_CHECK_INPUT(scales);
    GSX_CHECK_INPUT(colors);
    GSX_CHECK_INPUT(opacities);
    GSX_CHECK_INPUT(tile_offsets);
    GSX_CHECK_INPUT(flatten_ids);
    if (backgrounds.has_value()) { GSX_CHECK_INPUT(backgrounds.value()); }
    if (masks.has_value()) { GSX_CHECK_INPUT(masks.value()); }
    TORCH_CHECK(means.scalar_type() == at::kFloat && colors.scalar_type() == at::kFloat, "float32 only");
    const uint32_t C = tile_offsets.size(0), N = means.size(0);
    const uint32_t channels = colors.size(-1);
    if (channels != 3) {
        // The reference's --gut call site only passes 3 channels (rasterizer_autograd.cpp:285 stops its depth render modes), but the operator behind it
        // dispatches CDIM = 1, 2, 3, 4, 5, 8, ... (Rasterization.cpp:106-127; the assert(channels == 3) at :65 is compiled out in a Release build).
        // The blend is linear in the colours and its weights, alphas and last ids do not depend on them: any channel count = the 3-channel operator
        // on groups of three channels (the last group zero-padded).  The RGB hot path never comes through here.
        TORCH_CHECK(channels >= 1, "Unsupported number of channels: ", channels);
        TORCH_CHECK(g_fwd_ws_ready == nullptr && g_fwd_ws_out == nullptr && g_lists == nullptr, "the fused render path blends 3 channels");
        at::Tensor renders = at::empty({C, image_height, image_width, channels}, means.options()), alphas, last_ids;
        std::vector<int64_t> csz = colors.sizes().vec();
        csz.back() = 3;
        for (uint32_t c0 = 0; c0 < channels; c0 += 3) {
            const int64_t k = std::min<uint32_t>(3u, channels - c0);
            at::Tensor col3 = at::zeros(csz, colors.options());
            col3.narrow(-1, 0, k).copy_(colors.narrow(-1, c0, k));
            at::optional<at::Tensor> bg3;
            if (backgrounds.has_value()) {
                at::Tensor b = at::zeros({backgrounds->size(0), 3}, backgrounds->options());
                b.narrow(-1, 0, k).copy_(backgrounds->narrow(-1, c0, k));
                bg3 = b;
            }
            auto r = rasterize_to_pixels_from_world_3dgs_fwd(means, quats, scales, col3, opacities, bg3, masks, image_width, image_height, tile_size, viewmats0, viewmats1, Ks,
                                                             camera_model, ut_params, rs_type, radial_coeffs, tangential_coeffs, thin_prism_coeffs, tile_offsets, flatten_ids);
            renders.narrow(-1, c0, k).copy_(std::get<0>(r).narrow(-1, 0, k));
            if (c0 == 0) { alphas = std::get<1>(r); last_ids = std::get<2>(r); }
        }
        return std::make_tuple(renders, alphas, last_ids);
    }
    const gsx_cameras cams = make_cams(viewmats0, viewmats1, Ks, camera_model, rs_type, radial_coeffs, tangential_coeffs, thin_prism_coeffs, C);
    const gsx_ut_params ut = make_ut(ut_params);
    at::Tensor renders = at::empty({C, image_height, image_width, channels}, means.options());
    at::Tensor alphas = at::empty({C, image_height, image_width, 1}, means.options());
    at::Tensor last_ids = at::empty({C, image_height, image_width}, means.options().dtype(at::kInt));
    const size_t fwsb = gsx_rasterize_fwd_workspace_bytes(C, N);
    const bool ready = g_fwd_ws_ready && g_fwd_ws_ready->defined() && (size_t)g_fwd_ws_ready->numel() >= fwsb;   // packed by frontend_fused
    at::Tensor fws = ready ? *g_fwd_ws_ready : at::empty({(int64_t)fwsb}, means.options().dtype(at::kByte));
    // guarded lists (fused render path): flatten_ids has its capacity length, the kernels read the frame's verdict from the handle's device word
    const bool guarded = g_lists != nullptr && g_lists->status.defined();
    check(gsx_rasterize_to_pixels_from_world_3dgs_fwd_guarded(
              N, flatten_ids.size(0), means.data_ptr<float>(), quats.data_ptr<float>(), scales.data_ptr<float>(),
              colors.data_ptr<float>(), channels, opacities.data_ptr<float>(), fptr(backgrounds), bptr(masks), image_width,
              image_height, tile_size, &cams, &ut, tile_offsets.data_ptr<int32_t>(),
              flatten_ids.numel() ? flatten_ids.data_ptr<int32_t>() : nullptr, renders.data_ptr<float>(),
              alphas.data_ptr<float>(), last_ids.data_ptr<int32_t>(), fws.data_ptr(), (size_t)fws.numel(), ready ? 1 : 0,
              guarded ? g_lists->status.data_ptr<int32_t>() : nullptr, guarded ? g_lists->expected : 0, cur_stream()),
          "rasterize_to_pixels_from_world_3dgs_fwd");
    if (g_fwd_ws_out) *g_fwd_ws_out = fws;
    else if (!ready && shim_cache_on()) {   // the plain Ops.h call: keep the records for the backward of the same inputs (PackCache)
        std::lock_guard<std::mutex> lock(g_pack_cache_mutex);
        PackCache& pc = g_pack_cache;
        pc.means.set(means); pc.quats.set(quats); pc.scales.set(scales); pc.colors.set(colors); pc.opacities.set(opacities);
        pc.viewmats0.set(viewmats0); pc.Ks.set(Ks);
        pc.fws = fws; pc.W = image_width; pc.H = image_height; pc.C = C; pc.N = N; pc.camera_model = (int)camera_model; pc.shutter = (int)rs_type;
        auto given = [](const at::optional<at::Tensor>& o) { return o.has_value() && o->defined() && o->numel() > 0; };   // (the reference passes empty tensors for "none")
        pc.distorted = given(radial_coeffs) || given(tangential_coeffs) || given(thin_prism_coeffs) || given(viewmats1);
    }
    return std::make_tuple(renders, alphas, last_ids);
}

std::tuple<at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor> rasterize_to_pixels_from_world_3dgs_bwd(
    const at::Tensor means, const at::Tensor quats, const at::Tensor scales, const at::Tensor colors,
    const at::Tensor opacities, const at::optional<at::Tensor> backgrounds, const at::optional<at::Tensor> masks,
    const uint32_t image_width, const uint32_t image_height, const uint32_t tile_size, const at::Tensor viewmats0,
    const at::optional<at::Tensor> viewmats1, const at::Tensor Ks, const CameraModelType camera_model,
    const UnscentedTransformParameters ut_params, ShutterType rs_type, const at::optional<at::Tensor> radial_coeffs,
    const at::optional<at::Tensor> tangential_coeffs, const at::optional<at::Tensor> thin_prism_coeffs,
    const at::Tensor tile_offsets, const at::Tensor flatten_ids, const at::Tensor render_alphas,
    const at::Tensor last_ids, const at::Tensor v_render_colors, const at::Tensor v_render_alphas) {
    GSX_DEVICE_GUARD(means);
    GSX_CHECK_INPUT(means);
    GSX_CHECK_INPUT(quats);
    GSX_CHECK_INPUT(scales);
    GSX_CHECK_INPUT(colors);
    GSX_CHECK_INPUT(opacities);
    GSX_CHECK_INPUT(tile_offsets);
    GSX_CHECK_INPUT(flatten_ids);
    GSX_CHECK_INPUT(render_alphas);
    GSX_CHECK_INPUT(last_ids);
    GSX_CHECK_INPUT(v_render_colors);
    if (v_render_alphas.defined()) { GSX_CHECK_INPUT(v_render_alphas); }  // undefined = no gradient through the alpha output
    if (backgrounds.has_value()) { GSX_CHECK_INPUT(backgrounds.value()); }
    if (masks.has_value()) { GSX_CHECK_INPUT(masks.value()); }
    const uint32_t C = tile_offsets.size(0), N = means.size(0);
    const uint32_t channels = colors.size(-1);
    if (channels != 3) {
        // (see the forward) the loss is a sum over the channels: the gradients of the shared tensors are the sums over the channel groups — the alpha
        // output's gradient rides on the first group only —, v_colors is per channel
        TORCH_CHECK(channels >= 1, "Unsupported number of channels: ", channels);
        TORCH_CHECK(g_fwd_ws_in == nullptr && g_lists == nullptr && g_act == nullptr, "the fused render path blends 3 channels");
        at::Tensor v_means = at::zeros_like(means), v_quats = at::zeros_like(quats), v_scales = at::zeros_like(scales), v_opacities = at::zeros_like(opacities);
        at::Tensor v_colors = at::empty_like(colors);
        std::vector<int64_t> csz = colors.sizes().vec();
        csz.back() = 3;
        for (uint32_t c0 = 0; c0 < channels; c0 += 3) {
            const int64_t k = std::min<uint32_t>(3u, channels - c0);
            at::Tensor col3 = at::zeros(csz, colors.options());
            col3.narrow(-1, 0, k).copy_(colors.narrow(-1, c0, k));
            at::optional<at::Tensor> bg3;
            if (backgrounds.has_value()) {
                at::Tensor b = at::zeros({backgrounds->size(0), 3}, backgrounds->options());
                b.narrow(-1, 0, k).copy_(backgrounds->narrow(-1, c0, k));
                bg3 = b;
            }
            at::Tensor v3 = at::zeros({C, image_height, image_width, 3}, v_render_colors.options());
            v3.narrow(-1, 0, k).copy_(v_render_colors.narrow(-1, c0, k));
            auto r = rasterize_to_pixels_from_world_3dgs_bwd(means, quats, scales, col3, opacities, bg3, masks, image_width, image_height, tile_size, viewmats0, viewmats1, Ks,
                                                             camera_model, ut_params, rs_type, radial_coeffs, tangential_coeffs, thin_prism_coeffs, tile_offsets, flatten_ids,
                                                             render_alphas, last_ids, v3, c0 == 0 ? v_render_alphas : at::Tensor());
            v_means += std::get<0>(r); v_quats += std::get<1>(r); v_scales += std::get<2>(r); v_opacities += std::get<4>(r);
            v_colors.narrow(-1, c0, k).copy_(std::get<3>(r).narrow(-1, 0, k));
        }
        return std::make_tuple(v_means, v_quats, v_scales, v_colors, v_opacities);
    }
    const gsx_cameras cams = make_cams(viewmats0, viewmats1, Ks, camera_model, rs_type, radial_coeffs, tangential_coeffs, thin_prism_coeffs, C);
    const gsx_ut_params ut = make_ut(ut_params);
    at::Tensor v_means = at::empty_like(means);  // the C ABI overwrites all five (zero-fills itself where it scatters)
    at::Tensor v_quats = at::empty_like(quats);
    at::Tensor v_scales = at::empty_like(scales);
    at::Tensor v_colors = at::empty_like(colors);
    at::Tensor v_opacities = at::empty_like(opacities);
    // The workspace holds 64 B per intersection: 1.5 .. 1.8 GB at S-5M @4K, a different size for every camera.  Requests of ever-changing
    // sizes make the caching allocator split and re-grow its largest blocks (each growth is a hipMalloc of gigabytes inside the training
    // step), so the request is sized by the running maximum of n_isects for this problem shape — the same size call after call.
    int64_t isects_cap = flatten_ids.size(0);
    {
        static std::mutex cap_mutex;
        static std::map<std::tuple<int, uint32_t, uint32_t, uint32_t, uint32_t>, int64_t> caps;
        std::lock_guard<std::mutex> lock(cap_mutex);
        if (caps.size() > 4096) caps.clear();
        int64_t& cap = caps[std::make_tuple((int)means.get_device(), C, N, image_width, image_height)];
        if (isects_cap > cap || isects_cap < cap / 4) cap = isects_cap + isects_cap / 8;   // grows with 12 % head room; a much lighter scene starts over
        isects_cap = cap;
    }
    const size_t wsb = gsx_rasterize_bwd_workspace_bytes(C, N, tile_size == 32 ? 4 * isects_cap : isects_cap);   // lists per 32 x 32 pixels: four record slots per entry
    at::Tensor ws = at::empty({(int64_t)wsb}, means.options().dtype(at::kByte));  // caching allocator, like CUB temp storage upstream
    const void* packed = (g_fwd_ws_in && g_fwd_ws_in->defined())
                             ? gsx_rasterize_fwd_packed_records(g_fwd_ws_in->data_ptr(), (size_t)g_fwd_ws_in->numel(), C, N) : nullptr;
    at::Tensor cached_ws;   // (keeps the workspace alive across the launch even if another forward replaces the cache entry)
    if (packed == nullptr && g_fwd_ws_in == nullptr) {
        std::lock_guard<std::mutex> lock(g_pack_cache_mutex);
        PackCache& pc = g_pack_cache;
        auto given = [](const at::optional<at::Tensor>& o) { return o.has_value() && o->defined() && o->numel() > 0; };
        const bool distorted = given(radial_coeffs) || given(tangential_coeffs) || given(thin_prism_coeffs) || given(viewmats1);
        // (distorted / rolling-shutter cameras re-pack: their records depend on more tensors than the key holds; the reference's --gut training camera is a plain pinhole)
        if (pc.fws.defined() && !distorted && !pc.distorted && pc.C == C && pc.N == N && pc.W == image_width && pc.H == image_height && pc.camera_model == (int)camera_model &&
            pc.shutter == (int)rs_type && pc.means.matches(means) && pc.quats.matches(quats) && pc.scales.matches(scales) && pc.colors.matches(colors) &&
            pc.opacities.matches(opacities) && pc.viewmats0.matches(viewmats0) && pc.Ks.matches(Ks)) {
            cached_ws = std::move(pc.fws);   // handed out once: the workspace's record-chain heads are consumed and restored by THIS backward's gather — a second
            pc.reset();                      // backward of the same forward (retain_graph, another thread) packs for itself instead of sharing them
            packed = gsx_rasterize_fwd_packed_records(cached_ws.data_ptr(), (size_t)cached_ws.numel(), C, N);
        }
    }
    const bool guarded = g_lists != nullptr && g_lists->status.defined();
    if (g_act != nullptr) {   // through to the raw parameters (v_quats / v_scales / v_opacities are scratch: returned, not meaningful)
        check(gsx_rasterize_to_pixels_from_world_3dgs_bwd_act(
                  N, flatten_ids.size(0), means.data_ptr<float>(), quats.data_ptr<float>(), scales.data_ptr<float>(),
                  colors.data_ptr<float>(), channels, opacities.data_ptr<float>(), fptr(backgrounds), bptr(masks), image_width,
                  image_height, tile_size, &cams, &ut, tile_offsets.data_ptr<int32_t>(),
                  flatten_ids.numel() ? flatten_ids.data_ptr<int32_t>() : nullptr, render_alphas.data_ptr<float>(),
                  last_ids.data_ptr<int32_t>(), v_render_colors.data_ptr<float>(),
                  v_render_alphas.defined() ? v_render_alphas.data_ptr<float>() : nullptr,
                  v_means.data_ptr<float>(), v_quats.data_ptr<float>(), v_scales.data_ptr<float>(), v_colors.data_ptr<float>(),
                  v_opacities.data_ptr<float>(), ws.data_ptr(), wsb, packed, guarded ? g_lists->status.data_ptr<int32_t>() : nullptr,
                  guarded ? g_lists->expected : 0, g_act->scaling_raw.data_ptr<float>(), g_act->rotation_raw.data_ptr<float>(),
                  g_act->opacity_raw.data_ptr<float>(), g_act->v_scaling_raw.data_ptr<float>(), g_act->v_rotation_raw.data_ptr<float>(),
                  g_act->v_opacity_raw.data_ptr<float>(), g_act->scale_reg, g_act->opacity_reg, cur_stream()),
              "rasterize_to_pixels_from_world_3dgs_bwd(act)");
        return std::make_tuple(v_means, v_quats, v_scales, v_colors, v_opacities);
    }
    check(gsx_rasterize_to_pixels_from_world_3dgs_bwd_guarded(
              N, flatten_ids.size(0), means.data_ptr<float>(), quats.data_ptr<float>(), scales.data_ptr<float>(),
              colors.data_ptr<float>(), channels, opacities.data_ptr<float>(), fptr(backgrounds), bptr(masks), image_width,
              image_height, tile_size, &cams, &ut, tile_offsets.data_ptr<int32_t>(),
              flatten_ids.numel() ? flatten_ids.data_ptr<int32_t>() : nullptr, render_alphas.data_ptr<float>(),
              last_ids.data_ptr<int32_t>(), v_render_colors.data_ptr<float>(),
              v_render_alphas.defined() ? v_render_alphas.data_ptr<float>() : nullptr,
              v_means.data_ptr<float>(), v_quats.data_ptr<float>(), v_scales.data_ptr<float>(), v_colors.data_ptr<float>(),
              v_opacities.data_ptr<float>(), ws.data_ptr(), wsb, packed, guarded ? g_lists->status.data_ptr<int32_t>() : nullptr,
              guarded ? g_lists->expected : 0, cur_stream()),
          "rasterize_to_pixels_from_world_3dgs_bwd");
    return std::make_tuple(v_means, v_quats, v_scales, v_colors, v_opacities);
}

at::Tensor quats_to_rotmats(const at::Tensor quats) {
    GSX_DEVICE_GUARD(quats);
    GSX_CHECK_INPUT(quats);
    const uint32_t N = quats.size(0);
    at::Tensor rotmats = at::empty({N, 3, 3}, quats.options());
    check(gsx_quats_to_rotmats(N, quats.data_ptr<float>(), rotmats.data_ptr<float>(), cur_stream()), "quats_to_rotmats");
    return rotmats;
}

std::tuple<at::Tensor, at::Tensor> relocation(at::Tensor opacities, at::Tensor scales, at::Tensor ratios, at::Tensor binoms,
                                              const int n_max) {
    GSX_DEVICE_GUARD(opacities);
    GSX_CHECK_INPUT(opacities);
    GSX_CHECK_INPUT(scales);
    GSX_CHECK_INPUT(ratios);
    GSX_CHECK_INPUT(binoms);
    TORCH_CHECK(ratios.scalar_type() == at::kInt, "ratios must be int32");
    at::Tensor new_opacities = at::empty_like(opacities);
    at::Tensor new_scales = at::empty_like(scales);
    check(gsx_relocation(opacities.size(0), opacities.data_ptr<float>(), scales.data_ptr<float>(), ratios.data_ptr<int32_t>(),
                         binoms.data_ptr<float>(), n_max, new_opacities.data_ptr<float>(), new_scales.data_ptr<float>(), cur_stream()),
          "relocation");
    return std::make_tuple(new_opacities, new_scales);
}

void add_noise(at::Tensor raw_opacities, at::Tensor raw_scales, at::Tensor raw_quats, at::Tensor noise, at::Tensor means,
               const float current_lr) {
    GSX_DEVICE_GUARD(raw_opacities);
    GSX_CHECK_INPUT(raw_opacities);
    GSX_CHECK_INPUT(raw_scales);
    GSX_CHECK_INPUT(raw_quats);
    GSX_CHECK_INPUT(noise);
    GSX_CHECK_INPUT(means);
    check(gsx_add_noise(raw_opacities.size(0), raw_opacities.data_ptr<float>(), raw_scales.data_ptr<float>(), raw_quats.data_ptr<float>(),
                        noise.data_ptr<float>(), means.data_ptr<float>(), current_lr, cur_stream()), "add_noise");
}

}  // namespace gsplat


// ---------------------------------------------------------------------------------------------
// Fused glue ops (extensions beyond gsplat/Ops.h; see include/gsx.h "fused glue")
// ---------------------------------------------------------------------------------------------
namespace gsx_ext {

// colors [C,N,3] = clamp_min(SH(means - campos, coeffs | radii > 0) + 0.5, 0); masked rows are zero
at::Tensor sh_colors_fwd(const uint32_t degrees_to_use, const at::Tensor means, const at::Tensor viewmats,
                         const at::Tensor coeffs, const at::Tensor radii) {
    GSX_DEVICE_GUARD(means);
    GSX_CHECK_INPUT(means); GSX_CHECK_INPUT(viewmats); GSX_CHECK_INPUT(coeffs); GSX_CHECK_INPUT(radii);
    TORCH_CHECK(means.scalar_type() == at::kFloat && coeffs.scalar_type() == at::kFloat && radii.scalar_type() == at::kInt, "dtype");
    const uint32_t C = viewmats.size(0), N = means.size(0), K = coeffs.size(-2);
    at::Tensor colors = at::empty({C, N, 3}, means.options());  // the kernel writes every row (zeros where masked)
    check(gsx_sh_colors_fwd(degrees_to_use, C, N, K, means.data_ptr<float>(), viewmats.data_ptr<float>(), coeffs.data_ptr<float>(),
                            radii.data_ptr<int32_t>(), colors.data_ptr<float>(), cur_stream()), "sh_colors_fwd");
    return colors;
}

// writes v_coeffs (into `v_coeffs_out` if given) and v_means_out = v_means_in + d colors/d means (into `v_means_out` if given)
// radii / colors absent: v_colors [C,N,3] are pre-masked (include/gsx.h), the colour-exchange mode of the multi-GPU step
std::tuple<at::Tensor, at::Tensor> sh_colors_bwd(const uint32_t degrees_to_use, const at::Tensor means, const at::Tensor viewmats,
                                                 const at::Tensor coeffs, const at::optional<at::Tensor> radii, const at::optional<at::Tensor> colors,
                                                 const at::Tensor v_colors, const at::optional<at::Tensor> v_means_in,
                                                 const at::optional<at::Tensor> v_coeffs_out, const at::optional<at::Tensor> v_means_out) {
    GSX_DEVICE_GUARD(means);
    GSX_CHECK_INPUT(means); GSX_CHECK_INPUT(viewmats); GSX_CHECK_INPUT(coeffs); GSX_CHECK_INPUT(v_colors);
    const bool premasked = !(radii.has_value() && radii->defined());
    TORCH_CHECK(premasked == !(colors.has_value() && colors->defined()), "sh_colors_bwd: radii and colors are given together, or neither");
    if (!premasked) { GSX_CHECK_INPUT(radii.value()); GSX_CHECK_INPUT(colors.value()); }
    const uint32_t C = viewmats.size(0), N = means.size(0), K = coeffs.size(-2);
    TORCH_CHECK(v_colors.numel() == (int64_t)C * N * 3, "sh_colors_bwd: v_colors must be [C,N,3]");
    at::Tensor vc = (v_coeffs_out.has_value() && v_coeffs_out->defined()) ? v_coeffs_out.value() : at::empty_like(coeffs);
    at::Tensor vm = (v_means_out.has_value() && v_means_out->defined()) ? v_means_out.value() : at::empty_like(means);
    GSX_CHECK_INPUT(vc); GSX_CHECK_INPUT(vm);
    TORCH_CHECK(vc.numel() == coeffs.numel() && vm.numel() == means.numel(), "gradient sink has the wrong size");
    const float* vmi = nullptr;
    if (v_means_in.has_value() && v_means_in->defined()) { GSX_CHECK_INPUT(v_means_in.value()); vmi = v_means_in->data_ptr<float>(); }
    check(gsx_sh_colors_bwd(degrees_to_use, C, N, K, means.data_ptr<float>(), viewmats.data_ptr<float>(), coeffs.data_ptr<float>(),
                            premasked ? nullptr : radii->data_ptr<int32_t>(), premasked ? nullptr : colors->data_ptr<float>(), v_colors.data_ptr<float>(), vc.data_ptr<float>(), vmi,
                            vm.data_ptr<float>(), cur_stream()), "sh_colors_bwd");
    return std::make_tuple(vc, vm);
}

// SH backward + Adam step of the SH tensor in one launch (include/gsx.h): coeffs / exp_avg / exp_avg_sq are updated in place; returns v_means
at::Tensor sh_colors_bwd_adam(const uint32_t degrees_to_use, const at::Tensor means, const at::Tensor viewmats, at::Tensor coeffs,
                              const at::optional<at::Tensor> radii, const at::optional<at::Tensor> colors, const at::Tensor v_colors,
                              const at::optional<at::Tensor> v_means_in, const at::optional<at::Tensor> v_means_out, at::Tensor exp_avg,
                              at::Tensor exp_avg_sq, double step_sh0, double step_shN, bool do_sh0, bool do_shN, double beta1, double beta2,
                              double eps, double bc2_sqrt_rcp) {
    GSX_DEVICE_GUARD(means);
    GSX_CHECK_INPUT(means); GSX_CHECK_INPUT(viewmats); GSX_CHECK_INPUT(coeffs); GSX_CHECK_INPUT(v_colors); GSX_CHECK_INPUT(exp_avg); GSX_CHECK_INPUT(exp_avg_sq);
    const bool premasked = !(radii.has_value() && radii->defined());
    TORCH_CHECK(premasked == !(colors.has_value() && colors->defined()), "sh_colors_bwd_adam: radii and colors are given together, or neither");
    if (!premasked) { GSX_CHECK_INPUT(radii.value()); GSX_CHECK_INPUT(colors.value()); }
    const uint32_t C = viewmats.size(0), N = means.size(0), K = coeffs.size(-2);
    TORCH_CHECK(v_colors.numel() == (int64_t)C * N * 3, "sh_colors_bwd_adam: v_colors must be [C,N,3]");
    TORCH_CHECK(exp_avg.numel() == coeffs.numel() && exp_avg_sq.numel() == coeffs.numel(), "sh_colors_bwd_adam: moments must match coeffs");
    at::Tensor vm = (v_means_out.has_value() && v_means_out->defined()) ? v_means_out.value() : at::empty_like(means);
    GSX_CHECK_INPUT(vm);
    const float* vmi = nullptr;
    if (v_means_in.has_value() && v_means_in->defined()) { GSX_CHECK_INPUT(v_means_in.value()); vmi = v_means_in->data_ptr<float>(); }
    check(gsx_sh_colors_bwd_adam(degrees_to_use, C, N, K, means.data_ptr<float>(), viewmats.data_ptr<float>(), coeffs.data_ptr<float>(),
                                 premasked ? nullptr : radii->data_ptr<int32_t>(), premasked ? nullptr : colors->data_ptr<float>(),
                                 v_colors.data_ptr<float>(), vmi, vm.data_ptr<float>(), exp_avg.data_ptr<float>(), exp_avg_sq.data_ptr<float>(),
                                 (float)step_sh0, (float)step_shN, do_sh0 ? 1 : 0, do_shN ? 1 : 0, (float)beta1, (float)beta2, (float)eps,
                                 (float)bc2_sqrt_rcp, cur_stream()), "sh_colors_bwd_adam");
    return vm;
}

// activations + UT projection of ONE camera in one launch (include/gsx.h): returns scales, quats, opacities, radii, means2d, depths, conics
std::tuple<at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor> splat_activations_projection_ut(
    const at::Tensor means, const at::Tensor scaling_raw, const at::Tensor rotation_raw, const at::Tensor opacity_raw, const at::Tensor viewmats0,
    const at::Tensor Ks, const uint32_t image_width, const uint32_t image_height, const float eps2d, const float near_plane, const float far_plane,
    const float radius_clip, const gsplat::CameraModelType camera_model, const UnscentedTransformParameters ut_params,
    const at::optional<at::Tensor> radial_coeffs, const at::optional<at::Tensor> tangential_coeffs, const at::optional<at::Tensor> thin_prism_coeffs) {
    GSX_DEVICE_GUARD(means);
    GSX_CHECK_INPUT(means); GSX_CHECK_INPUT(scaling_raw); GSX_CHECK_INPUT(rotation_raw); GSX_CHECK_INPUT(opacity_raw);
    TORCH_CHECK(means.scalar_type() == at::kFloat, "float32 only");
    const uint32_t N = means.size(0), C = Ks.size(0);
    TORCH_CHECK(C == 1, "splat_activations_projection_ut: one camera per call");
    const gsx_cameras cams = make_cams(viewmats0, at::nullopt, Ks, camera_model, ShutterType::GLOBAL, radial_coeffs, tangential_coeffs, thin_prism_coeffs, C);
    const gsx_ut_params ut = make_ut(ut_params);
    at::Tensor scales = at::empty_like(scaling_raw), quats = at::empty_like(rotation_raw);
    at::Tensor opac = at::empty({(int64_t)N}, means.options());
    at::Tensor radii = at::empty({C, N, 2}, means.options().dtype(at::kInt));
    at::Tensor means2d = at::empty({C, N, 2}, means.options());
    at::Tensor depths = at::empty({C, N}, means.options());
    at::Tensor conics = at::empty({C, N, 3}, means.options());
    check(gsx_splat_activations_projection_ut(N, means.data_ptr<float>(), rotation_raw.data_ptr<float>(), scaling_raw.data_ptr<float>(),
                                              opacity_raw.data_ptr<float>(), &cams, image_width, image_height, eps2d, near_plane, far_plane,
                                              radius_clip, &ut, scales.data_ptr<float>(), quats.data_ptr<float>(), opac.data_ptr<float>(),
                                              radii.data_ptr<int32_t>(), means2d.data_ptr<float>(), depths.data_ptr<float>(),
                                              conics.data_ptr<float>(), cur_stream()), "splat_activations_projection_ut");
    return std::make_tuple(scales, quats, opac, radii, means2d, depths, conics);
}

// The whole per-Gaussian front end of one render in one launch (include/gsx.h: gsx_frontend_fused): returns scales, quats, opacities,
// radii, means2d, depths, conics, colors and the blend-forward workspace with the packed records; an undefined workspace = not
// supported for these arguments (the caller runs the separate operators).
std::tuple<at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor> frontend_fused(
    const uint32_t degrees_to_use, const at::Tensor means, const at::Tensor sh, const at::Tensor scaling_raw, const at::Tensor rotation_raw,
    const at::Tensor opacity_raw, const at::Tensor viewmats0, const at::Tensor Ks, const uint32_t image_width, const uint32_t image_height,
    const float eps2d, const float near_plane, const float far_plane, const float radius_clip, const gsplat::CameraModelType camera_model,
    const UnscentedTransformParameters ut_params, const at::optional<at::Tensor> radial_coeffs, const at::optional<at::Tensor> tangential_coeffs,
    const at::optional<at::Tensor> thin_prism_coeffs, const bool want_conics, const bool record_ranges = false) {
    GSX_DEVICE_GUARD(means);
    GSX_CHECK_INPUT(means); GSX_CHECK_INPUT(sh); GSX_CHECK_INPUT(scaling_raw); GSX_CHECK_INPUT(rotation_raw); GSX_CHECK_INPUT(opacity_raw);
    TORCH_CHECK(means.scalar_type() == at::kFloat && sh.scalar_type() == at::kFloat, "float32 only");
    TORCH_CHECK(sh.dim() == 3 && sh.size(0) == means.size(0) && sh.size(2) == 3, "sh must be [N,K,3]");
    const uint32_t N = means.size(0), C = Ks.size(0), K = sh.size(1);
    const gsx_cameras cams = make_cams(viewmats0, at::nullopt, Ks, camera_model, ShutterType::GLOBAL, radial_coeffs, tangential_coeffs, thin_prism_coeffs, C);
    const gsx_ut_params ut = make_ut(ut_params);
    at::Tensor none;
    if (N == 0 || !gsx_frontend_fused_supported(K, degrees_to_use, &cams, sh.data_ptr<float>()))
        return std::make_tuple(none, none, none, none, none, none, none, none, none);
    at::Tensor scales = at::empty_like(scaling_raw), quats = at::empty_like(rotation_raw);
    at::Tensor opac = at::empty({(int64_t)N}, means.options());
    at::Tensor radii = at::empty({C, N, 2}, means.options().dtype(at::kInt));
    at::Tensor means2d = at::empty({C, N, 2}, means.options());
    at::Tensor depths = at::empty({C, N}, means.options());
    at::Tensor conics = want_conics ? at::empty({C, N, 3}, means.options()) : at::empty({0}, means.options());   // (nothing on the render path reads them)
    at::Tensor colors = at::empty({C, N, 3}, means.options());
    const size_t fwsb = gsx_rasterize_fwd_workspace_bytes(C, N);
    at::Tensor fws = at::empty({(int64_t)fwsb}, means.options().dtype(at::kByte));
    check(gsx_frontend_fused(N, K, degrees_to_use, means.data_ptr<float>(), rotation_raw.data_ptr<float>(), scaling_raw.data_ptr<float>(),
                             opacity_raw.data_ptr<float>(), sh.data_ptr<float>(), &cams, image_width, image_height, eps2d, near_plane, far_plane,
                             radius_clip, &ut, scales.data_ptr<float>(), quats.data_ptr<float>(), opac.data_ptr<float>(), radii.data_ptr<int32_t>(),
                             means2d.data_ptr<float>(), depths.data_ptr<float>(), want_conics ? conics.data_ptr<float>() : nullptr, colors.data_ptr<float>(), fws.data_ptr(),
                             fwsb, record_ranges ? 1 : 0, cur_stream()), "frontend_fused");
    return std::make_tuple(scales, quats, opac, radii, means2d, depths, conics, colors, fws);
}

std::tuple<at::Tensor, at::Tensor, at::Tensor> splat_activations_fwd(const at::Tensor scaling_raw, const at::Tensor rotation_raw,
                                                                     const at::Tensor opacity_raw) {
    GSX_DEVICE_GUARD(scaling_raw);
    GSX_CHECK_INPUT(scaling_raw); GSX_CHECK_INPUT(rotation_raw); GSX_CHECK_INPUT(opacity_raw);
    const uint32_t N = scaling_raw.size(0);
    at::Tensor scales = at::empty_like(scaling_raw), quats = at::empty_like(rotation_raw);
    at::Tensor opac = at::empty({(int64_t)N}, scaling_raw.options());
    check(gsx_splat_activations_fwd(N, scaling_raw.data_ptr<float>(), rotation_raw.data_ptr<float>(), opacity_raw.data_ptr<float>(),
                                    scales.data_ptr<float>(), quats.data_ptr<float>(), opac.data_ptr<float>(), cur_stream()), "splat_activations_fwd");
    return std::make_tuple(scales, quats, opac);
}

std::tuple<at::Tensor, at::Tensor, at::Tensor> splat_activations_bwd(const at::Tensor scaling_raw, const at::Tensor rotation_raw,
                                                                     const at::Tensor opacity_raw, const at::Tensor v_scales,
                                                                     const at::Tensor v_quats, const at::Tensor v_opacities,
                                                                     const at::optional<at::Tensor> out_scaling,
                                                                     const at::optional<at::Tensor> out_rotation,
                                                                     const at::optional<at::Tensor> out_opacity, const double scale_reg_per_element,
                                                                     const double opacity_reg_per_element) {
    GSX_DEVICE_GUARD(scaling_raw);
    GSX_CHECK_INPUT(scaling_raw); GSX_CHECK_INPUT(rotation_raw); GSX_CHECK_INPUT(opacity_raw);
    GSX_CHECK_INPUT(v_scales); GSX_CHECK_INPUT(v_quats); GSX_CHECK_INPUT(v_opacities);
    const uint32_t N = scaling_raw.size(0);
    at::Tensor gs = (out_scaling.has_value() && out_scaling->defined()) ? out_scaling.value() : at::empty_like(scaling_raw);
    at::Tensor gr = (out_rotation.has_value() && out_rotation->defined()) ? out_rotation.value() : at::empty_like(rotation_raw);
    at::Tensor go = (out_opacity.has_value() && out_opacity->defined()) ? out_opacity.value() : at::empty_like(opacity_raw);
    GSX_CHECK_INPUT(gs); GSX_CHECK_INPUT(gr); GSX_CHECK_INPUT(go);
    check(gsx_splat_activations_bwd_reg(N, scaling_raw.data_ptr<float>(), rotation_raw.data_ptr<float>(), opacity_raw.data_ptr<float>(),
                                        v_scales.data_ptr<float>(), v_quats.data_ptr<float>(), v_opacities.data_ptr<float>(),
                                        gs.data_ptr<float>(), gr.data_ptr<float>(), go.data_ptr<float>(), (float)scale_reg_per_element,
                                        (float)opacity_reg_per_element, cur_stream()), "splat_activations_bwd");
    return std::make_tuple(gs, gr, go);
}

void adam_step_split(at::Tensor param, at::Tensor exp_avg, at::Tensor exp_avg_sq, const at::Tensor grad, int64_t split, double lr_a, double lr_b,
                     bool step_a, bool step_b, double beta1, double beta2, double eps, double bias_correction1_rcp, double bias_correction2_sqrt_rcp) {
    GSX_DEVICE_GUARD(param);
    TORCH_CHECK(param.is_cuda() && grad.is_cuda() && exp_avg.is_cuda() && exp_avg_sq.is_cuda(), "adam_step_split: CUDA tensors required");
    TORCH_CHECK(param.is_contiguous() && grad.is_contiguous() && exp_avg.is_contiguous() && exp_avg_sq.is_contiguous(), "adam_step_split: dense tensors required");
    TORCH_CHECK(param.sizes() == grad.sizes() && param.sizes() == exp_avg.sizes() && param.sizes() == exp_avg_sq.sizes() && param.dim() >= 2,
                "adam_step_split: shape mismatch");
    if (param.numel() == 0) return;
    const uint64_t rows = param.size(0);
    const uint32_t cols = (uint32_t)(param.numel() / param.size(0));
    check(gsx_adam_step_split(rows, cols, (uint32_t)split, param.data_ptr<float>(), exp_avg.data_ptr<float>(), exp_avg_sq.data_ptr<float>(),
                              grad.data_ptr<float>(), (float)lr_a, (float)lr_b, step_a, step_b, (float)beta1, (float)beta2, (float)eps,
                              (float)bias_correction1_rcp, (float)bias_correction2_sqrt_rcp, cur_stream()), "adam_step_split");
}

// intersect_tile(sort = true) + intersect_offset through the binned pipeline: (tiles_per_gauss, isect_ids | empty, flatten_ids,
// isect_offsets [C, tile_height, tile_width]); same values as the two reference ops.
// `lists` == nullptr: the exact protocol — the op's outputs have exactly n_isects rows, so the host has to read that number (the one
// sync of the op, as upstream: Intersect.cpp:76).  To keep the GPU busy meanwhile, the fill is launched optimistically into buffers
// sized from the capacity hint; the host then waits only for the 8-byte copy, not for the fill, and repeats the fill if the guess was
// too small.
// `lists` != nullptr: the guarded protocol (include/gsx.h) — nothing blocks: flatten_ids keeps its capacity length, the verdict lands in
// lists->status on the device and in lists->n_host on the host, lists->confirm() reads it later.  Without a hint (first call of a
// problem shape) the exact protocol runs and the handle comes back confirmed.
static std::tuple<at::Tensor, at::Tensor, at::Tensor, at::Tensor> intersect_tile_binned_core(const at::Tensor& means2d, const at::Tensor& radii,
                                                                                             const at::Tensor& depths, const uint32_t C,
                                                                                             const uint32_t tile_size, const uint32_t tile_width,
                                                                                             const uint32_t tile_height, const bool want_isect_ids,
                                                                                             IsectLists* lists) {
    GSX_DEVICE_GUARD(means2d);
    GSX_CHECK_INPUT(means2d); GSX_CHECK_INPUT(radii); GSX_CHECK_INPUT(depths);
    TORCH_CHECK(means2d.dim() == 3, "intersect_tile_binned: means2d must be [C,N,2]");
    TORCH_CHECK(means2d.scalar_type() == at::kFloat && depths.scalar_type() == at::kFloat && radii.scalar_type() == at::kInt, "dtype");
    if (!gsx_intersect_bin_supported(tile_width, tile_height)) {  // tile grid too large for LDS counters: the device-wide sort
        auto r = gsplat::intersect_tile(means2d, radii, depths, at::nullopt, at::nullopt, C, tile_size, tile_width, tile_height, true);
        at::Tensor off = gsplat::intersect_offset(std::get<1>(r), C, tile_width, tile_height);
        if (lists) { lists->confirmed = true; lists->n_isects = std::get<2>(r).size(0); lists->expected = lists->capacity = lists->n_isects; }
        return std::make_tuple(std::get<0>(r), want_isect_ids ? std::get<1>(r) : at::empty({0}, std::get<1>(r).options()), std::get<2>(r), off);
    }
    const uint32_t n_elements = means2d.numel() / 2, N = C ? n_elements / C : 0;
    void* st = cur_stream();
    const HintKey key = std::make_tuple((int)means2d.get_device(), C, N, tile_width, tile_height);
    int64_t hint = 0, hint_seg = 0, last_total = 0;
    hint_lookup(key, hint, hint_seg, &last_total);
    // Frames with heavy tiles (last call: a tile above 4096 keys and at least kRankedMeanKeys keys per tile on average) take the ranked
    // fill: one frame-wide depth ranking (~50 us), then 4-byte keys and bitmap sorts for the heavy tiles.  Same outputs bit for bit,
    // so the choice — like the capacity hint — only affects speed.  GSX_INTERSECT_FILL=ranked|keys forces one or the other.
    constexpr int64_t kRankedMeanKeys = 2500;
    const int64_t nseg_all = (int64_t)C * tile_width * tile_height;
    bool ranked = hint_seg > 4096 && hint >= kRankedMeanKeys * nseg_all;
    if (const char* e = gsx_test_switch("GSX_INTERSECT_FILL")) ranked = std::strcmp(e, "ranked") == 0 ? true : (std::strcmp(e, "keys") == 0 ? false : ranked);
    ranked = ranked && n_elements && gsx_intersect_ranked_supported(C, N);
    int64_t capacity = 0, seg_bound = 0;
    if (hint > 0 && n_elements) {
        capacity = std::min<int64_t>(hint + hint / 4 + 4096, 0x7FFFFFFFll);   // 25 % head room: 288 GB of HBM make slots cheaper than repeats
        // bound on the largest (camera, tile) segment the fill is launched for, in the tiers of its sort kernels: <= 1024 keys: the one-wave
        // sort alone; <= 4096: + the block sort; <= 16384: + the 1024-thread sort; above: + giant-segment merge passes for exactly this bound
        const int64_t sb = hint_seg + hint_seg / 4;
        seg_bound = sb <= 1024 ? 1024 : (sb <= 4096 ? 4096 : std::max<int64_t>(sb, 16384));
    }
    const bool guarded = lists != nullptr && capacity > 0;

    // (the guarded entry's callers never look at tiles_per_gauss: the count kernel then skips that 4 B / Gaussian store)
    at::Tensor tiles_per_gauss = lists ? at::empty({0}, depths.options().dtype(at::kInt)) : at::empty_like(depths, depths.options().dtype(at::kInt));
    at::Tensor offsets = at::empty({(int64_t)C * tile_height * tile_width + 1}, depths.options().dtype(at::kInt));
    const size_t cwb = gsx_intersect_bin_count_workspace_bytes(C, tile_width, tile_height);
    at::Tensor cws = at::empty({(int64_t)cwb}, depths.options().dtype(at::kByte));
    const HostWord n_host = HostWord::take();
    at::Tensor status;
    if (guarded) status = at::empty({1}, depths.options().dtype(at::kInt));
    check(gsx_intersect_bin_count_guarded(C, N, n_elements ? means2d.data_ptr<float>() : nullptr, n_elements ? radii.data_ptr<int32_t>() : nullptr, tile_size,
                                          tile_width, tile_height, (n_elements && !lists) ? tiles_per_gauss.data_ptr<int32_t>() : nullptr, offsets.data_ptr<int32_t>(),
                                          (int64_t*)n_host.p, cws.data_ptr(), cwb, capacity, ranked ? 0 : seg_bound,
                                          guarded ? status.data_ptr<int32_t>() : nullptr, st), "intersect_tile_binned(count)");
    if (const char* e = gsx_test_switch("GSX_COUNT_EVENT"); e && e[0] == '1') {   // A/B tool: what rounds 2 .. 4 did here — an event behind the count
        at::cuda::CUDAEvent ev;
        ev.record(c10::hip::getCurrentHIPStreamMasqueradingAsCUDA());
    }
    at::Tensor ranks, order;
    if (ranked) {
        ranks = at::empty({(int64_t)n_elements}, depths.options().dtype(at::kInt));
        order = at::empty({(int64_t)n_elements}, depths.options().dtype(at::kInt));
        const size_t rwb = gsx_intersect_depth_ranks_workspace_bytes(C, N);
        at::Tensor rws = at::empty({(int64_t)rwb}, depths.options().dtype(at::kByte));
        check(gsx_intersect_depth_ranks(C, N, radii.data_ptr<int32_t>(), depths.data_ptr<float>(), (uint32_t*)ranks.data_ptr<int32_t>(),
                                        (uint32_t*)order.data_ptr<int32_t>(), rws.data_ptr(), rwb, st), "intersect_tile_binned(ranks)");
        g_stats.ranked_calls++;
    }
    at::Tensor flatten_ids, isect_ids;
    auto fill = [&](int64_t cap, int64_t bound) {
        flatten_ids = at::empty({cap}, depths.options().dtype(at::kInt));
        isect_ids = at::empty({want_isect_ids ? cap : 0}, depths.options().dtype(at::kLong));
        if (ranked) {
            const size_t fwb = gsx_intersect_bin_fill_ranked_workspace_bytes(cap);
            at::Tensor fws = at::empty({(int64_t)fwb}, depths.options().dtype(at::kByte));
            check(gsx_intersect_bin_fill_ranked(C, N, means2d.data_ptr<float>(), radii.data_ptr<int32_t>(), depths.data_ptr<float>(), tile_size,
                                                tile_width, tile_height, offsets.data_ptr<int32_t>(), cap, cws.data_ptr(),
                                                (const uint32_t*)ranks.data_ptr<int32_t>(), (const uint32_t*)order.data_ptr<int32_t>(),
                                                flatten_ids.data_ptr<int32_t>(), want_isect_ids ? isect_ids.data_ptr<int64_t>() : nullptr,
                                                fws.data_ptr(), fwb, st), "intersect_tile_binned(ranked fill)");
            return;
        }
        const size_t fwb = gsx_intersect_bin_fill_workspace_bytes(C, tile_width, tile_height, cap);
        at::Tensor fws = at::empty({(int64_t)fwb}, depths.options().dtype(at::kByte));
        check(gsx_intersect_bin_fill(C, N, means2d.data_ptr<float>(), radii.data_ptr<int32_t>(), depths.data_ptr<float>(), tile_size, tile_width,
                                     tile_height, offsets.data_ptr<int32_t>(), cap, bound, cws.data_ptr(), flatten_ids.data_ptr<int32_t>(),
                                     want_isect_ids ? isect_ids.data_ptr<int64_t>() : nullptr, fws.data_ptr(), fwb, st), "intersect_tile_binned(fill)");
    };
    if (capacity > 0) fill(capacity, seg_bound);
    at::Tensor isect_offsets = offsets.narrow(0, 0, (int64_t)C * tile_height * tile_width).view({(int64_t)C, (int64_t)tile_height, (int64_t)tile_width});
    g_stats.binned_calls++;
    if (guarded) {   // nobody waits: the consumers read the verdict on the device, the host confirms later
        lists->n_host = n_host; lists->status = status; lists->key = key;
        lists->capacity = capacity; lists->seg_bound = seg_bound; lists->ranked = ranked;
        lists->expected = last_total > 0 ? std::min(last_total, capacity) : capacity;
        g_stats.guarded_calls++;
        return std::make_tuple(tiles_per_gauss, isect_ids, flatten_ids, isect_offsets);
    }
    const uint64_t word = n_host.wait();   // the exact protocol: the host waits for the count here, every call
    g_stats.host_syncs++;
    const int64_t n_isects = (int64_t)(word & 0xFFFFFFFFull), max_seg = (int64_t)(word >> 32);
    TORCH_CHECK(n_isects <= 0x7FFFFFFFll, "intersect_tile: more than 2^31 - 1 intersections (tile offsets are int32, as upstream's isect_offsets)");
    hint_update(key, n_isects, max_seg);
    const bool seg_ok = ranked || max_seg <= seg_bound;   // (the ranked fill has no merge passes to run short of)
    if (capacity > 0 && (n_isects > capacity || !seg_ok)) g_stats.hint_misses++;
    if (capacity == 0 && n_isects > 0) g_stats.hint_cold++;
    if (capacity > 0 && n_isects <= capacity && seg_ok) {
        flatten_ids = flatten_ids.narrow(0, 0, n_isects);
        if (want_isect_ids) isect_ids = isect_ids.narrow(0, 0, n_isects);
    } else if (n_isects > 0) {
        fill(n_isects, std::max<int64_t>(max_seg, 1));
    } else {
        flatten_ids = at::empty({0}, depths.options().dtype(at::kInt));
        isect_ids = at::empty({0}, depths.options().dtype(at::kLong));
    }
    if (lists) {   // cold call of the guarded entry: exact lists, already confirmed
        lists->confirmed = true; lists->complete = true; lists->n_isects = n_isects; lists->max_seg = max_seg;
        lists->capacity = lists->expected = n_isects; lists->key = key;
    }
    return std::make_tuple(tiles_per_gauss, isect_ids, flatten_ids, isect_offsets);
}

std::tuple<at::Tensor, at::Tensor, at::Tensor, at::Tensor> intersect_tile_binned(const at::Tensor means2d, const at::Tensor radii,
                                                                                 const at::Tensor depths, const uint32_t C,
                                                                                 const uint32_t tile_size, const uint32_t tile_width,
                                                                                 const uint32_t tile_height, const bool want_isect_ids) {
    return intersect_tile_binned_core(means2d, radii, depths, C, tile_size, tile_width, tile_height, want_isect_ids, nullptr);
}

// the guarded protocol: (tiles_per_gauss [empty: not produced], flatten_ids [capacity], isect_offsets, handle)
std::tuple<at::Tensor, at::Tensor, at::Tensor, std::shared_ptr<IsectLists>> intersect_tile_binned_guarded(const at::Tensor means2d, const at::Tensor radii,
                                                                                                          const at::Tensor depths, const uint32_t C,
                                                                                                          const uint32_t tile_size, const uint32_t tile_width,
                                                                                                          const uint32_t tile_height) {
    auto lists = std::make_shared<IsectLists>();
    auto r = intersect_tile_binned_core(means2d, radii, depths, C, tile_size, tile_width, tile_height, false, lists.get());
    return std::make_tuple(std::get<0>(r), std::get<2>(r), std::get<3>(r), lists);
}

// fusedssim / fusedssim_backward (include/kernels/ssim.cuh:11-29): same tuple returns
std::tuple<at::Tensor, at::Tensor, at::Tensor, at::Tensor> fusedssim(double C1, double C2, const at::Tensor& img1_, const at::Tensor& img2_, bool train) {
    GSX_DEVICE_GUARD(img1_);
    TORCH_CHECK(img1_.is_cuda() && img2_.is_cuda() && img1_.dim() == 4 && img1_.sizes() == img2_.sizes(), "fusedssim: two [B,CH,H,W] CUDA tensors required");
    TORCH_CHECK(img1_.scalar_type() == at::kFloat && img2_.scalar_type() == at::kFloat, "fusedssim: float32 required");
    const at::Tensor img1 = img1_.contiguous(), img2 = img2_.contiguous();
    at::Tensor map = at::empty_like(img1);
    at::Tensor d0 = train ? at::empty_like(img1) : at::empty({0}, img1.options());
    at::Tensor d1 = train ? at::empty_like(img1) : at::empty({0}, img1.options());
    at::Tensor d2 = train ? at::empty_like(img1) : at::empty({0}, img1.options());
    check(gsx_fused_ssim_fwd((uint32_t)img1.size(0), (uint32_t)img1.size(1), (uint32_t)img1.size(2), (uint32_t)img1.size(3), (float)C1, (float)C2,
                             img1.data_ptr<float>(), img2.data_ptr<float>(), map.data_ptr<float>(), train ? d0.data_ptr<float>() : nullptr,
                             train ? d1.data_ptr<float>() : nullptr, train ? d2.data_ptr<float>() : nullptr, cur_stream()), "fusedssim");
    return std::make_tuple(map, d0, d1, d2);
}

at::Tensor fusedssim_backward(double C1, double C2, const at::Tensor& img1_, const at::Tensor& img2_, const at::Tensor& dL_dmap_,
                              const at::Tensor& dm_dmu1, const at::Tensor& dm_dsigma1_sq, const at::Tensor& dm_dsigma12) {
    GSX_DEVICE_GUARD(img1_);
    TORCH_CHECK(img1_.is_cuda() && img1_.dim() == 4 && img1_.sizes() == img2_.sizes() && img1_.sizes() == dL_dmap_.sizes(), "fusedssim_backward: shape mismatch");
    TORCH_CHECK(dm_dmu1.sizes() == img1_.sizes() && dm_dsigma1_sq.sizes() == img1_.sizes() && dm_dsigma12.sizes() == img1_.sizes(),
                "fusedssim_backward: derivative maps of the forward (train = true) required");
    const at::Tensor img1 = img1_.contiguous(), img2 = img2_.contiguous(), dL = dL_dmap_.contiguous();
    const at::Tensor a = dm_dmu1.contiguous(), b = dm_dsigma1_sq.contiguous(), c = dm_dsigma12.contiguous();
    at::Tensor out = at::empty_like(img1);
    check(gsx_fused_ssim_bwd((uint32_t)img1.size(0), (uint32_t)img1.size(1), (uint32_t)img1.size(2), (uint32_t)img1.size(3), (float)C1, (float)C2,
                             img1.data_ptr<float>(), img2.data_ptr<float>(), dL.data_ptr<float>(), out.data_ptr<float>(), a.data_ptr<float>(),
                             b.data_ptr<float>(), c.data_ptr<float>(), cur_stream()), "fusedssim_backward");
    return out;
}

// fused photometric loss on the blend's [C,H,W,3] output; returns (loss3 = {loss, l1, ssim}, workspace for the backward)
std::tuple<at::Tensor, at::Tensor> photometric_loss_fwd(const at::Tensor& render, const at::Tensor& gt, double lambda_dssim) {
    GSX_DEVICE_GUARD(render);
    TORCH_CHECK(render.is_cuda() && gt.is_cuda() && render.dim() == 4 && render.size(3) == 3 && render.is_contiguous(), "photometric_loss: render must be contiguous [C,H,W,3]");
    TORCH_CHECK(gt.dim() == 4 && gt.size(0) == render.size(0) && gt.size(1) == 3 && gt.size(2) == render.size(1) && gt.size(3) == render.size(2) &&
                    gt.is_contiguous(), "photometric_loss: gt must be contiguous [C,3,H,W]");
    TORCH_CHECK(render.scalar_type() == at::kFloat && gt.scalar_type() == at::kFloat, "photometric_loss: float32 required");
    const uint32_t C = (uint32_t)render.size(0), H = (uint32_t)render.size(1), W = (uint32_t)render.size(2);
    const size_t bytes = gsx_photometric_loss_workspace_bytes(C, H, W);
    at::Tensor ws = at::empty({(int64_t)bytes}, render.options().dtype(at::kByte));
    at::Tensor loss3 = at::empty({3}, render.options());
    check(gsx_photometric_loss_fwd(C, H, W, (float)lambda_dssim, render.data_ptr<float>(), gt.data_ptr<float>(), loss3.data_ptr<float>(),
                                   ws.data_ptr(), bytes, cur_stream()), "photometric_loss_fwd");
    return std::make_tuple(loss3, ws);
}

// the training loss and its gradient in one kernel (include/gsx.h ABI 7): (loss3, v_render = grad_scale * d loss / d render)
std::tuple<at::Tensor, at::Tensor> photometric_loss_single_pass(const at::Tensor& render, const at::Tensor& gt, double lambda_dssim, double grad_scale) {
    GSX_DEVICE_GUARD(render);
    TORCH_CHECK(render.is_cuda() && gt.is_cuda() && render.dim() == 4 && render.size(3) == 3 && render.is_contiguous(), "photometric_loss: render must be contiguous [C,H,W,3]");
    TORCH_CHECK(gt.dim() == 4 && gt.size(0) == render.size(0) && gt.size(1) == 3 && gt.size(2) == render.size(1) && gt.size(3) == render.size(2) &&
                    gt.is_contiguous(), "photometric_loss: gt must be contiguous [C,3,H,W]");
    TORCH_CHECK(render.scalar_type() == at::kFloat && gt.scalar_type() == at::kFloat, "photometric_loss: float32 required");
    const uint32_t C = (uint32_t)render.size(0), H = (uint32_t)render.size(1), W = (uint32_t)render.size(2);
    const size_t bytes = gsx_photometric_loss_single_pass_workspace_bytes(C, H, W);
    at::Tensor ws = at::empty({(int64_t)bytes}, render.options().dtype(at::kByte));
    at::Tensor loss3 = at::empty({3}, render.options());
    at::Tensor v = at::empty_like(render);
    check(gsx_photometric_loss_single_pass(C, H, W, (float)lambda_dssim, (float)grad_scale, render.data_ptr<float>(), gt.data_ptr<float>(), loss3.data_ptr<float>(),
                                           v.data_ptr<float>(), ws.data_ptr(), bytes, cur_stream()), "photometric_loss_single_pass");
    return std::make_tuple(loss3, v);
}

at::Tensor photometric_loss_bwd(const at::Tensor& render, const at::Tensor& gt, const at::Tensor& ws, double lambda_dssim,
                                const c10::optional<at::Tensor>& grad_loss, double grad_scale) {
    GSX_DEVICE_GUARD(render);
    const uint32_t C = (uint32_t)render.size(0), H = (uint32_t)render.size(1), W = (uint32_t)render.size(2);
    at::Tensor v = at::empty_like(render);
    const float* gl = nullptr;
    at::Tensor glt;
    if (grad_loss.has_value() && grad_loss->defined()) {
        glt = grad_loss->to(at::kFloat).contiguous();
        TORCH_CHECK(glt.is_cuda() && glt.numel() == 1, "photometric_loss_bwd: grad_loss must be a device scalar");
        gl = glt.data_ptr<float>();
    }
    check(gsx_photometric_loss_bwd(C, H, W, (float)lambda_dssim, gl, (float)grad_scale, render.data_ptr<float>(), gt.data_ptr<float>(), ws.data_ptr(),
                                   (size_t)ws.numel(), v.data_ptr<float>(), cur_stream()), "photometric_loss_bwd");
    return v;
}

// fused Adam step on a parameter (or a row-strided view of one: dims after the first must be dense)
// several dense parameter groups in one launch (include/gsx.h: gsx_adam_step_multi); lrs / bias corrections per tensor
void adam_step_multi(std::vector<at::Tensor> params, std::vector<at::Tensor> exp_avgs, std::vector<at::Tensor> exp_avg_sqs,
                     std::vector<at::Tensor> grads, std::vector<double> lrs, std::vector<double> bc1_rcps, std::vector<double> bc2_sqrt_rcps,
                     double beta1, double beta2, double eps) {
    const size_t k = params.size();
    TORCH_CHECK(k > 0 && k <= GSX_ADAM_MULTI_MAX && exp_avgs.size() == k && exp_avg_sqs.size() == k && grads.size() == k && lrs.size() == k &&
                bc1_rcps.size() == k && bc2_sqrt_rcps.size() == k, "adam_step_multi: 1..", GSX_ADAM_MULTI_MAX, " tensors, equally long lists");
    GSX_DEVICE_GUARD(params[0]);
    float* p[GSX_ADAM_MULTI_MAX]; float* m[GSX_ADAM_MULTI_MAX]; float* v[GSX_ADAM_MULTI_MAX]; const float* g[GSX_ADAM_MULTI_MAX];
    uint64_t n[GSX_ADAM_MULTI_MAX]; float lr[GSX_ADAM_MULTI_MAX], b1[GSX_ADAM_MULTI_MAX], b2[GSX_ADAM_MULTI_MAX];
    for (size_t i = 0; i < k; ++i) {
        GSX_CHECK_INPUT(params[i]); GSX_CHECK_INPUT(exp_avgs[i]); GSX_CHECK_INPUT(exp_avg_sqs[i]); GSX_CHECK_INPUT(grads[i]);
        TORCH_CHECK(params[i].numel() == grads[i].numel() && params[i].numel() == exp_avgs[i].numel() && params[i].numel() == exp_avg_sqs[i].numel(),
                    "adam_step_multi: shape mismatch in tensor ", i);
        p[i] = params[i].data_ptr<float>(); m[i] = exp_avgs[i].data_ptr<float>(); v[i] = exp_avg_sqs[i].data_ptr<float>(); g[i] = grads[i].data_ptr<float>();
        n[i] = (uint64_t)params[i].numel(); lr[i] = (float)lrs[i]; b1[i] = (float)bc1_rcps[i]; b2[i] = (float)bc2_sqrt_rcps[i];
    }
    check(gsx_adam_step_multi((uint32_t)k, p, m, v, g, n, lr, b1, b2, (float)beta1, (float)beta2, (float)eps, cur_stream()), "adam_step_multi");
}

void adam_step(at::Tensor param, at::Tensor exp_avg, at::Tensor exp_avg_sq, const at::Tensor grad, double lr, double beta1,
               double beta2, double eps, double bias_correction1_rcp, double bias_correction2_sqrt_rcp) {
    GSX_DEVICE_GUARD(param);
    TORCH_CHECK(param.is_cuda() && grad.is_cuda() && exp_avg.is_cuda() && exp_avg_sq.is_cuda(), "adam_step: CUDA tensors required");
    TORCH_CHECK(exp_avg.is_contiguous() && exp_avg_sq.is_contiguous(), "adam_step: optimizer states must be contiguous");
    TORCH_CHECK(param.sizes() == grad.sizes() && param.numel() == exp_avg.numel() && param.numel() == exp_avg_sq.numel(), "adam_step: shape mismatch");
    if (param.numel() == 0) return;
    auto rows_cols_ld = [](const at::Tensor& t, uint64_t& rows, uint32_t& cols, uint64_t& ld) {
        if (t.is_contiguous()) { rows = 1; cols = 0; ld = 0; return true; }
        if (t.dim() < 2) return false;
        int64_t inner = 1;
        for (int64_t d = t.dim() - 1; d >= 1; --d) { if (t.stride(d) != inner) return false; inner *= t.size(d); }
        rows = t.size(0); cols = (uint32_t)inner; ld = t.stride(0);
        return true;
    };
    uint64_t rp, rg, lp, lg; uint32_t cp, cg;
    TORCH_CHECK(rows_cols_ld(param, rp, cp, lp) && rows_cols_ld(grad, rg, cg, lg), "adam_step: only dense or row-strided tensors are supported");
    uint64_t rows; uint32_t cols;
    if (cp == 0 && cg == 0) { rows = 1; cols = 0; }
    else { rows = param.size(0); cols = (uint32_t)(param.numel() / param.size(0)); }
    if (cols == 0) {  // both dense: treat as one long row, chunked to fit uint32 columns
        const uint64_t n = param.numel();
        TORCH_CHECK(n < (1ull << 32), "adam_step: dense tensors above 2^32 elements are not supported");
        rows = 1; cols = (uint32_t)n; lp = lg = n;
    } else {
        if (cp == 0) lp = cols;
        if (cg == 0) lg = cols;
    }
    check(gsx_adam_step(rows, cols, lp, lg, param.data_ptr<float>(), exp_avg.data_ptr<float>(), exp_avg_sq.data_ptr<float>(),
                        grad.data_ptr<float>(), (float)lr, (float)beta1, (float)beta2, (float)eps, (float)bias_correction1_rcp,
                        (float)bias_correction2_sqrt_rcp, cur_stream()), "adam_step");
}

}  // namespace gsx_ext

// ---------------------------------------------------------------------------------------------
// Link-level drop-ins for the reference's Adam and SSIM operators (include/gsx_training_ops.h)
// ---------------------------------------------------------------------------------------------
namespace fast_gs::optimizer {

void adam_step_wrapper(at::Tensor& param, at::Tensor& exp_avg, at::Tensor& exp_avg_sq, const at::Tensor& param_grad, const float lr,
                       const float beta1, const float beta2, const float eps, const float bias_correction1_rcp,
                       const float bias_correction2_sqrt_rcp) {
    gsx_ext::adam_step(param, exp_avg, exp_avg_sq, param_grad, lr, beta1, beta2, eps, bias_correction1_rcp, bias_correction2_sqrt_rcp);
}

void adam_step(float* param, float* exp_avg, float* exp_avg_sq, const float* param_grad, const int n_elements, const float lr,
               const float beta1, const float beta2, const float eps, const float bias_correction1_rcp,
               const float bias_correction2_sqrt_rcp) {
    if (n_elements <= 0) return;
    check(gsx_adam_step(1, (uint32_t)n_elements, (uint64_t)n_elements, (uint64_t)n_elements, param, exp_avg, exp_avg_sq, param_grad, lr, beta1,
                        beta2, eps, bias_correction1_rcp, bias_correction2_sqrt_rcp, cur_stream()), "adam_step");
}

}  // namespace fast_gs::optimizer

std::tuple<at::Tensor, at::Tensor, at::Tensor, at::Tensor> fusedssim(float C1, float C2, at::Tensor& img1, at::Tensor& img2, bool train) {
    return gsx_ext::fusedssim(C1, C2, img1, img2, train);
}

at::Tensor fusedssim_backward(float C1, float C2, at::Tensor& img1, at::Tensor& img2, at::Tensor& dL_dmap, at::Tensor& dm_dmu1,
                              at::Tensor& dm_dsigma1_sq, at::Tensor& dm_dsigma12) {
    return gsx_ext::fusedssim_backward(C1, C2, img1, img2, dL_dmap, dm_dmu1, dm_dsigma1_sq, dm_dsigma12);
}

// ---------------------------------------------------------------------------------------------
// Python bindings (names as in gsplat/Ops.h)
// ---------------------------------------------------------------------------------------------
#ifndef GSX_NO_PYBIND
namespace py = pybind11;

// (renders, alphas, last_ids) -> (renders, alphas, last_ids, workspace)
template <class... A>
static auto keep_fwd_ws(std::tuple<at::Tensor, at::Tensor, at::Tensor> (*fn)(A...)) {
    return [fn](A... a) {
        at::Tensor ws;
        struct Reset { ~Reset() { g_fwd_ws_out = nullptr; } } reset;
        g_fwd_ws_out = &ws;
        auto r = fn(a...);
        return std::make_tuple(std::get<0>(r), std::get<1>(r), std::get<2>(r), ws);
    };
}

PYBIND11_MODULE(_gsx_ops, m) {
    m.doc() = "gsplat operator surface on the MI355X HIP backend (libgsx.so)";
    py::class_<UnscentedTransformParameters>(m, "UnscentedTransformParameters")
        .def(py::init<>())
        .def_readwrite("alpha", &UnscentedTransformParameters::alpha)
        .def_readwrite("beta", &UnscentedTransformParameters::beta)
        .def_readwrite("kappa", &UnscentedTransformParameters::kappa)
        .def_readwrite("in_image_margin_factor", &UnscentedTransformParameters::in_image_margin_factor)
        .def_readwrite("require_all_sigma_points_valid", &UnscentedTransformParameters::require_all_sigma_points_valid);
    py::enum_<gsplat::CameraModelType>(m, "CameraModelType")
        .value("PINHOLE", gsplat::PINHOLE).value("ORTHO", gsplat::ORTHO).value("FISHEYE", gsplat::FISHEYE);
    py::enum_<ShutterType>(m, "ShutterType")
        .value("ROLLING_TOP_TO_BOTTOM", ShutterType::ROLLING_TOP_TO_BOTTOM)
        .value("ROLLING_LEFT_TO_RIGHT", ShutterType::ROLLING_LEFT_TO_RIGHT)
        .value("ROLLING_BOTTOM_TO_TOP", ShutterType::ROLLING_BOTTOM_TO_TOP)
        .value("ROLLING_RIGHT_TO_LEFT", ShutterType::ROLLING_RIGHT_TO_LEFT)
        .value("GLOBAL", ShutterType::GLOBAL);
    py::class_<IsectLists, std::shared_ptr<IsectLists>>(m, "IsectLists")
        .def_property_readonly("status", [](const IsectLists& l) { return l.status.defined() ? at::optional<at::Tensor>(l.status) : at::nullopt; })
        .def_readonly("capacity", &IsectLists::capacity)
        .def_readonly("expected", &IsectLists::expected)
        .def_readonly("confirmed", &IsectLists::confirmed)
        .def("ready", &IsectLists::is_ready)     // non-blocking: has the GPU passed the count?
        .def("confirm", &IsectLists::confirm);   // (n_isects, largest segment, complete); waits for the count if it has to
    m.def("spherical_harmonics_fwd", &gsplat::spherical_harmonics_fwd);
    m.def("spherical_harmonics_bwd", &gsplat::spherical_harmonics_bwd);
    m.def("intersect_tile", &gsplat::intersect_tile);
    m.def("intersect_offset", &gsplat::intersect_offset);
    m.def("projection_ut_3dgs_fused", &gsplat::projection_ut_3dgs_fused);
    m.def("rasterize_to_pixels_from_world_3dgs_fwd", &gsplat::rasterize_to_pixels_from_world_3dgs_fwd);
    m.def("rasterize_fwd_keep_ws", keep_fwd_ws(&gsplat::rasterize_to_pixels_from_world_3dgs_fwd));
    m.def("rasterize_fwd_packed",   // the blend forward on a workspace whose records frontend_fused already packed (same inputs)
          [](const at::Tensor means, const at::Tensor quats, const at::Tensor scales, const at::Tensor colors, const at::Tensor opacities,
             const at::optional<at::Tensor> backgrounds, const at::optional<at::Tensor> masks, uint32_t image_width, uint32_t image_height,
             uint32_t tile_size, const at::Tensor viewmats0, const at::optional<at::Tensor> viewmats1, const at::Tensor Ks,
             const gsplat::CameraModelType camera_model, const UnscentedTransformParameters ut_params, ShutterType rs_type,
             const at::optional<at::Tensor> radial_coeffs, const at::optional<at::Tensor> tangential_coeffs,
             const at::optional<at::Tensor> thin_prism_coeffs, const at::Tensor tile_offsets, const at::Tensor flatten_ids, const at::Tensor fwd_ws,
             std::shared_ptr<IsectLists> lists) {          // handle of intersect_tile_binned_guarded (None: exact lists)
              struct Reset { ~Reset() { g_fwd_ws_ready = nullptr; g_lists = nullptr; } } reset;
              g_fwd_ws_ready = &fwd_ws;
              g_lists = lists.get();
              return gsplat::rasterize_to_pixels_from_world_3dgs_fwd(means, quats, scales, colors, opacities, backgrounds, masks, image_width, image_height,
                                                                     tile_size, viewmats0, viewmats1, Ks, camera_model, ut_params, rs_type, radial_coeffs,
                                                                     tangential_coeffs, thin_prism_coeffs, tile_offsets, flatten_ids);
          }, py::arg("means"), py::arg("quats"), py::arg("scales"), py::arg("colors"), py::arg("opacities"), py::arg("backgrounds"), py::arg("masks"),
          py::arg("image_width"), py::arg("image_height"), py::arg("tile_size"), py::arg("viewmats0"), py::arg("viewmats1"), py::arg("Ks"),
          py::arg("camera_model"), py::arg("ut_params"), py::arg("rs_type"), py::arg("radial_coeffs"), py::arg("tangential_coeffs"),
          py::arg("thin_prism_coeffs"), py::arg("tile_offsets"), py::arg("flatten_ids"), py::arg("fwd_ws"), py::arg("lists") = std::shared_ptr<IsectLists>());
    // frontend_fused(…): all outputs; frontend_fused_render(…): the render path's call — `conics` (read by nothing downstream) comes back empty
    m.def("frontend_fused", [](uint32_t deg, at::Tensor means, at::Tensor sh, at::Tensor sr, at::Tensor rr, at::Tensor orw, at::Tensor vm, at::Tensor Ks,
                               uint32_t w, uint32_t h, float eps2d, float nearp, float farp, float clip, gsplat::CameraModelType cm,
                               UnscentedTransformParameters ut, at::optional<at::Tensor> rad, at::optional<at::Tensor> tang, at::optional<at::Tensor> prism) {
        return gsx_ext::frontend_fused(deg, means, sh, sr, rr, orw, vm, Ks, w, h, eps2d, nearp, farp, clip, cm, ut, rad, tang, prism, true);
    });
    m.def("frontend_fused_render", [](uint32_t deg, at::Tensor means, at::Tensor sh, at::Tensor sr, at::Tensor rr, at::Tensor orw, at::Tensor vm, at::Tensor Ks,
                                      uint32_t w, uint32_t h, float eps2d, float nearp, float farp, float clip, gsplat::CameraModelType cm,
                                      UnscentedTransformParameters ut, at::optional<at::Tensor> rad, at::optional<at::Tensor> tang, at::optional<at::Tensor> prism,
                                      bool record_ranges) {   // record_ranges: the backward's records in one contiguous run per Gaussian (frames of large footprints)
        return gsx_ext::frontend_fused(deg, means, sh, sr, rr, orw, vm, Ks, w, h, eps2d, nearp, farp, clip, cm, ut, rad, tang, prism, false, record_ranges);
    });
    m.def("rasterize_to_pixels_from_world_3dgs_bwd",
          [](const at::Tensor means, const at::Tensor quats, const at::Tensor scales, const at::Tensor colors, const at::Tensor opacities,
             const at::optional<at::Tensor> backgrounds, const at::optional<at::Tensor> masks, uint32_t image_width, uint32_t image_height,
             uint32_t tile_size, const at::Tensor viewmats0, const at::optional<at::Tensor> viewmats1, const at::Tensor Ks,
             const gsplat::CameraModelType camera_model, const UnscentedTransformParameters ut_params, ShutterType rs_type,
             const at::optional<at::Tensor> radial_coeffs, const at::optional<at::Tensor> tangential_coeffs,
             const at::optional<at::Tensor> thin_prism_coeffs, const at::Tensor tile_offsets, const at::Tensor flatten_ids,
             const at::Tensor render_alphas, const at::Tensor last_ids, const at::Tensor v_render_colors,
             const at::optional<at::Tensor> v_render_alphas,  // None = no gradient through the alpha output
             const at::optional<at::Tensor> fwd_ws,             // workspace kept from rasterize_fwd_keep_ws of the same inputs
             std::shared_ptr<IsectLists> lists) {               // handle of intersect_tile_binned_guarded the forward ran with (None: exact lists)
              struct Reset { ~Reset() { g_fwd_ws_in = nullptr; g_lists = nullptr; } } reset;
              g_fwd_ws_in = fwd_ws.has_value() ? &fwd_ws.value() : nullptr;
              g_lists = lists.get();
              return gsplat::rasterize_to_pixels_from_world_3dgs_bwd(
                  means, quats, scales, colors, opacities, backgrounds, masks, image_width, image_height, tile_size, viewmats0, viewmats1, Ks,
                  camera_model, ut_params, rs_type, radial_coeffs, tangential_coeffs, thin_prism_coeffs, tile_offsets, flatten_ids,
                  render_alphas, last_ids, v_render_colors, v_render_alphas.has_value() ? v_render_alphas.value() : at::Tensor());
          }, py::arg("means"), py::arg("quats"), py::arg("scales"), py::arg("colors"), py::arg("opacities"), py::arg("backgrounds"), py::arg("masks"),
          py::arg("image_width"), py::arg("image_height"), py::arg("tile_size"), py::arg("viewmats0"), py::arg("viewmats1"), py::arg("Ks"),
          py::arg("camera_model"), py::arg("ut_params"), py::arg("rs_type"), py::arg("radial_coeffs"), py::arg("tangential_coeffs"),
          py::arg("thin_prism_coeffs"), py::arg("tile_offsets"), py::arg("flatten_ids"), py::arg("render_alphas"), py::arg("last_ids"),
          py::arg("v_render_colors"), py::arg("v_render_alphas"), py::arg("fwd_ws") = at::optional<at::Tensor>(), py::arg("lists") = std::shared_ptr<IsectLists>());
    // The blend backward of ONE camera through to the raw SplatData parameters (include/gsx.h ABI 7: the activation Jacobians ride on the gather
    // kernel): returns (v_means, v_colors, v_scaling_raw, v_rotation_raw, v_opacity_raw); out_* = caller-provided gradient buffers (sinks).
    m.def("rasterize_bwd_act",
          [](const at::Tensor means, const at::Tensor quats, const at::Tensor scales, const at::Tensor colors, const at::Tensor opacities,
             const at::optional<at::Tensor> backgrounds, const at::optional<at::Tensor> masks, uint32_t image_width, uint32_t image_height,
             uint32_t tile_size, const at::Tensor viewmats0, const at::optional<at::Tensor> viewmats1, const at::Tensor Ks,
             const gsplat::CameraModelType camera_model, const UnscentedTransformParameters ut_params, ShutterType rs_type,
             const at::optional<at::Tensor> radial_coeffs, const at::optional<at::Tensor> tangential_coeffs,
             const at::optional<at::Tensor> thin_prism_coeffs, const at::Tensor tile_offsets, const at::Tensor flatten_ids,
             const at::Tensor render_alphas, const at::Tensor last_ids, const at::Tensor v_render_colors,
             const at::optional<at::Tensor> v_render_alphas, const at::optional<at::Tensor> fwd_ws, std::shared_ptr<IsectLists> lists,
             const at::Tensor scaling_raw, const at::Tensor rotation_raw, const at::Tensor opacity_raw, const at::optional<at::Tensor> out_scaling,
             const at::optional<at::Tensor> out_rotation, const at::optional<at::Tensor> out_opacity, double scale_reg_per_element,
             double opacity_reg_per_element) {
              GSX_CHECK_INPUT(scaling_raw); GSX_CHECK_INPUT(rotation_raw); GSX_CHECK_INPUT(opacity_raw);
              TORCH_CHECK(tile_offsets.size(0) == 1, "rasterize_bwd_act: one camera only");
              const int64_t N = means.size(0);
              TORCH_CHECK(scaling_raw.numel() == 3 * N && rotation_raw.numel() == 4 * N && opacity_raw.numel() == N, "rasterize_bwd_act: raw parameter shapes");
              ActArgs act;
              act.scaling_raw = scaling_raw; act.rotation_raw = rotation_raw; act.opacity_raw = opacity_raw;
              act.v_scaling_raw = (out_scaling.has_value() && out_scaling->defined()) ? out_scaling.value() : at::empty_like(scaling_raw);
              act.v_rotation_raw = (out_rotation.has_value() && out_rotation->defined()) ? out_rotation.value() : at::empty_like(rotation_raw);
              act.v_opacity_raw = (out_opacity.has_value() && out_opacity->defined()) ? out_opacity.value() : at::empty_like(opacity_raw);
              GSX_CHECK_INPUT(act.v_scaling_raw); GSX_CHECK_INPUT(act.v_rotation_raw); GSX_CHECK_INPUT(act.v_opacity_raw);
              TORCH_CHECK(act.v_scaling_raw.numel() == 3 * N && act.v_rotation_raw.numel() == 4 * N && act.v_opacity_raw.numel() == N, "rasterize_bwd_act: gradient buffer shapes");
              act.scale_reg = (float)scale_reg_per_element; act.opacity_reg = (float)opacity_reg_per_element;
              struct Reset { ~Reset() { g_fwd_ws_in = nullptr; g_lists = nullptr; g_act = nullptr; } } reset;
              g_fwd_ws_in = fwd_ws.has_value() ? &fwd_ws.value() : nullptr;
              g_lists = lists.get();
              g_act = &act;
              auto r = gsplat::rasterize_to_pixels_from_world_3dgs_bwd(
                  means, quats, scales, colors, opacities, backgrounds, masks, image_width, image_height, tile_size, viewmats0, viewmats1, Ks,
                  camera_model, ut_params, rs_type, radial_coeffs, tangential_coeffs, thin_prism_coeffs, tile_offsets, flatten_ids,
                  render_alphas, last_ids, v_render_colors, v_render_alphas.has_value() ? v_render_alphas.value() : at::Tensor());
              return std::make_tuple(std::get<0>(r), std::get<3>(r), act.v_scaling_raw, act.v_rotation_raw, act.v_opacity_raw);
          });
    m.def("quats_to_rotmats", &gsplat::quats_to_rotmats);
    m.def("relocation", &gsplat::relocation);
    m.def("add_noise", &gsplat::add_noise);
    m.def("abi_version", []() { return gsx_abi_version(); });
    m.def("shim_stats", [](bool reset) {  // (host_syncs, binned intersect calls, capacity-hint misses, cold calls without a hint)
        auto r = std::make_tuple((int64_t)g_stats.host_syncs, (int64_t)g_stats.binned_calls, (int64_t)g_stats.hint_misses, (int64_t)g_stats.hint_cold);
        if (reset) { g_stats.host_syncs = 0; g_stats.binned_calls = 0; g_stats.hint_misses = 0; g_stats.hint_cold = 0; }
        return r;
    });
    m.def("shim_ranked_calls", [](bool reset) {  // binned intersect calls that took the ranked fill (heavy tiles)
        const int64_t r = g_stats.ranked_calls;
        if (reset) g_stats.ranked_calls = 0;
        return r;
    });
    m.def("sh_colors_fwd", &gsx_ext::sh_colors_fwd);
    m.def("sh_colors_bwd", &gsx_ext::sh_colors_bwd);
    m.def("sh_colors_bwd_adam", &gsx_ext::sh_colors_bwd_adam);
    m.def("splat_activations_fwd", &gsx_ext::splat_activations_fwd);
    m.def("splat_activations_projection_ut", &gsx_ext::splat_activations_projection_ut);
    m.def("splat_activations_bwd", &gsx_ext::splat_activations_bwd, py::arg("scaling_raw"), py::arg("rotation_raw"), py::arg("opacity_raw"), py::arg("v_scales"),
          py::arg("v_quats"), py::arg("v_opacities"), py::arg("out_scaling") = at::optional<at::Tensor>(), py::arg("out_rotation") = at::optional<at::Tensor>(),
          py::arg("out_opacity") = at::optional<at::Tensor>(), py::arg("scale_reg_per_element") = 0.0, py::arg("opacity_reg_per_element") = 0.0);
    m.def("intersect_tile_binned", &gsx_ext::intersect_tile_binned);
    m.def("intersect_tile_binned_guarded", &gsx_ext::intersect_tile_binned_guarded);
    m.def("shim_guarded_stats", [](bool reset) {  // (guarded intersect calls, confirms that had to wait for the GPU, frames whose lists were incomplete)
        auto r = std::make_tuple((int64_t)g_stats.guarded_calls, (int64_t)g_stats.guarded_waits, (int64_t)g_stats.guarded_misses);
        if (reset) { g_stats.guarded_calls = 0; g_stats.guarded_waits = 0; g_stats.guarded_misses = 0; }
        return r;
    });
    m.def("intersect_tile_device_sort", [](const at::Tensor means2d, const at::Tensor radii, const at::Tensor depths, uint32_t C, uint32_t tile_size,
                                           uint32_t tile_width, uint32_t tile_height, bool sort) {
        return gsplat::intersect_tile_device_sort(means2d, radii, depths, C, tile_size, tile_width, tile_height, sort);
    });
    m.def("adam_step", &gsx_ext::adam_step);
    m.def("adam_step_multi", &gsx_ext::adam_step_multi);
    m.def("adam_step_wrapper", [](at::Tensor param, at::Tensor exp_avg, at::Tensor exp_avg_sq, const at::Tensor grad, float lr, float beta1, float beta2,
                                  float eps, float bc1_rcp, float bc2_sqrt_rcp) {  // the reference's signature (adam_api.h:11-21)
        fast_gs::optimizer::adam_step_wrapper(param, exp_avg, exp_avg_sq, grad, lr, beta1, beta2, eps, bc1_rcp, bc2_sqrt_rcp);
    });
    m.def("adam_step_split", &gsx_ext::adam_step_split);
    m.def("fusedssim", &gsx_ext::fusedssim);
    m.def("fusedssim_backward", &gsx_ext::fusedssim_backward);
    m.def("photometric_loss_fwd", &gsx_ext::photometric_loss_fwd);
    m.def("photometric_loss_bwd", &gsx_ext::photometric_loss_bwd);
    m.def("photometric_loss_single_pass", &gsx_ext::photometric_loss_single_pass);
}
#endif  // GSX_NO_PYBIND
