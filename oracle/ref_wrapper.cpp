// ref_wrapper.cpp — thin extern "C" shim OVER the reference's own tests/torch_impl.cpp.
// TEST INFRASTRUCTURE ONLY.  Nothing from the reference is copied: torch_impl.cpp is compiled from
// where it lies under /root/reference (see build_ref.sh); this file only marshals raw pointers
// into CPU torch tensors and calls reference::* (declared in /root/reference/tests/torch_impl.hpp).
#include <cstdint>
#include <cstring>
#include <torch/torch.h>

#include "torch_impl.hpp"

namespace {
torch::Tensor f32(const float* p, std::vector<int64_t> shape) {
    return torch::from_blob(const_cast<float*>(p), shape, torch::kFloat32).clone();
}
}  // namespace

extern "C" void ref_quat_to_rotmat(int64_t n, const float* quats, float* out) {
    auto R = reference::quat_to_rotmat(f32(quats, {n, 4})).contiguous();
    std::memcpy(out, R.data_ptr<float>(), sizeof(float) * n * 9);
}

// (covars, precis) = reference::quat_scale_to_covar_preci(quats, scales, true, true, false): full 3x3 matrices
extern "C" void ref_quat_scale_to_covar_preci(int64_t n, const float* quats, const float* scales, float* covars, float* precis) {
    auto [c, p] = reference::quat_scale_to_covar_preci(f32(quats, {n, 4}), f32(scales, {n, 3}), true, true, false);
    auto cc = c.contiguous(), pc = p.contiguous();
    std::memcpy(covars, cc.data_ptr<float>(), sizeof(float) * n * 9);
    std::memcpy(precis, pc.data_ptr<float>(), sizeof(float) * n * 9);
}

// colors = reference::spherical_harmonics(degree, dirs, coeffs); optional grads through torch autograd
extern "C" void ref_spherical_harmonics(int degree, int64_t n, int64_t K, const float* dirs, const float* coeffs,
                                        float* colors, const float* v_colors, float* v_coeffs, float* v_dirs) {
    auto d = f32(dirs, {n, 3});
    auto c = f32(coeffs, {n, K, 3});
    if (v_colors) {
        d.requires_grad_(true);
        c.requires_grad_(true);
    }
    auto col = reference::spherical_harmonics(degree, d, c);
    auto colc = col.detach().contiguous();
    std::memcpy(colors, colc.data_ptr<float>(), sizeof(float) * n * 3);
    if (v_colors) {
        auto loss = (col * f32(v_colors, {n, 3})).sum();
        auto grads = torch::autograd::grad({loss}, {c, d}, {}, true, false, true);
        auto gc = grads[0].defined() ? grads[0].contiguous() : torch::zeros({n, K, 3});
        auto gd = grads[1].defined() ? grads[1].contiguous() : torch::zeros({n, 3});
        std::memcpy(v_coeffs, gc.data_ptr<float>(), sizeof(float) * n * K * 3);
        std::memcpy(v_dirs, gd.data_ptr<float>(), sizeof(float) * n * 3);
    }
}

// returns n_isects; fills the output buffers when capacity suffices
extern "C" int64_t ref_isect_tiles(int64_t C, int64_t N, const float* means2d, const int32_t* radii, const float* depths,
                                   int tile_size, int tile_width, int tile_height, int sort, int32_t* tiles_per_gauss,
                                   int64_t capacity, int64_t* isect_ids, int32_t* flatten_ids) {
    auto m = f32(means2d, {C, N, 2});
    auto r = torch::from_blob(const_cast<int32_t*>(radii), {C, N, 2}, torch::kInt32).clone();
    auto d = f32(depths, {C, N});
    auto [tpg, ids, fl] = reference::isect_tiles(m, r, d, tile_size, tile_width, tile_height, sort != 0);
    auto tpgc = tpg.to(torch::kInt32).contiguous();
    std::memcpy(tiles_per_gauss, tpgc.data_ptr<int32_t>(), sizeof(int32_t) * C * N);
    int64_t n = ids.size(0);
    if (n <= capacity && n > 0) {
        auto idc = ids.to(torch::kInt64).contiguous();
        auto flc = fl.to(torch::kInt32).contiguous();
        std::memcpy(isect_ids, idc.data_ptr<int64_t>(), sizeof(int64_t) * n);
        std::memcpy(flatten_ids, flc.data_ptr<int32_t>(), sizeof(int32_t) * n);
    }
    return n;
}

// The EWA stages of torch_impl (NOT the UT projection): used only as the CPU-baseline "reference" leg
// and as a sanity bound for means2d.
extern "C" void ref_ewa_projection(int64_t C, int64_t N, const float* means, const float* quats, const float* scales,
                                   const float* viewmats, const float* Ks, int width, int height, float eps2d,
                                   float near_plane, float far_plane, int32_t* radii, float* means2d, float* depths,
                                   float* conics) {
    auto [covars, precis] = reference::quat_scale_to_covar_preci(f32(quats, {N, 4}), f32(scales, {N, 3}), true, false, false);
    auto [rad, m2d, dep, con, comp] = reference::fully_fused_projection(
        f32(means, {N, 3}), covars, f32(viewmats, {C, 4, 4}), f32(Ks, {C, 3, 3}), width, height, eps2d, near_plane,
        far_plane, false, "pinhole");
    auto rc = rad.to(torch::kInt32).contiguous();
    auto mc = m2d.contiguous(), dc = dep.contiguous(), cc = con.contiguous();
    std::memcpy(radii, rc.data_ptr<int32_t>(), sizeof(int32_t) * rc.numel());
    std::memcpy(means2d, mc.data_ptr<float>(), sizeof(float) * mc.numel());
    std::memcpy(depths, dc.data_ptr<float>(), sizeof(float) * dc.numel());
    std::memcpy(conics, cc.data_ptr<float>(), sizeof(float) * cc.numel());
}

extern "C" int ref_num_threads(void) { return at::get_num_threads(); }
