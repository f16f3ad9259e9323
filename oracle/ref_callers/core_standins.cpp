// TEST INFRASTRUCTURE ONLY (oracle/build_ref_callers.sh).  The members of the reference's gs::Camera / gs::SplatData / gs::geometry classes that
// its render call sites (src/training/rasterization/rasterizer.cpp, rasterizer_autograd.cpp — compiled UNMODIFIED) reference, defined
// here against the reference's OWN headers (include/core/camera.hpp, include/core/splat_data.hpp, include/geometry/*.hpp, read where they
// lie).  The reference defines them in src/core/camera.cpp / splat_data.cpp / src/geometry/*.cpp next to image IO, PLY / SOG export and
// logging (OpenImageIO, tinyply, spdlog, TBB ...: not in this image); what is restated here is the arithmetic of the few that matter to a
// render — each cites the lines it follows — and trivial bodies for the rest.  Never linked into the product.
#include "core/camera.hpp"
#include "core/splat_data.hpp"
#include "geometry/bounding_box.hpp"

using torch::indexing::None;
using torch::indexing::Slice;

namespace gs {
    // [R | t; 0 0 0 1] as a [1,4,4] float32 tensor on the GPU: what src/core/camera.cpp:15-23 builds with index_put_
    static torch::Tensor rigid_w2c(const torch::Tensor& R, const torch::Tensor& t) {
        const auto f32 = torch::TensorOptions().dtype(torch::kFloat32).device(R.device());
        const auto upper = torch::cat({R.to(f32), t.to(f32).reshape({3, 1})}, 1);
        const auto last_row = torch::tensor({0.f, 0.f, 0.f, 1.f}, f32).reshape({1, 4});
        return torch::cat({upper, last_row}, 0).to(torch::kCUDA).unsqueeze(0).contiguous();
    }

    // the members src/core/camera.cpp:25-58 initialises (declaration order of include/core/camera.hpp), image size = camera size
    Camera::Camera(const torch::Tensor& R, const torch::Tensor& T, float focal_x, float focal_y, float center_x, float center_y,
                   const torch::Tensor radial_distortion, const torch::Tensor tangential_distortion, gsplat::CameraModelType camera_model_type,
                   const std::string& image_name, const std::filesystem::path& image_path, int camera_width, int camera_height, int uid)
        : _uid(uid), _focal_x(focal_x), _focal_y(focal_y), _center_x(center_x), _center_y(center_y), _R(R), _T(T),
          _radial_distortion(radial_distortion), _tangential_distortion(tangential_distortion), _camera_model_type(camera_model_type),
          _image_name(image_name), _image_path(image_path), _camera_width(camera_width), _camera_height(camera_height),
          _image_width(camera_width), _image_height(camera_height), _world_view_transform{rigid_w2c(R, T)} {
        _cam_position = torch::inverse(_world_view_transform[0]).index({Slice(None, 3), 3}).contiguous();
        _FoVx = focal2fov(_focal_x, _camera_width);
        _FoVy = focal2fov(_focal_y, _camera_height);
    }

    // [1,3,3] intrinsics of the (possibly resized) image on the view matrix's device — the values of src/core/camera.cpp:82-103
    std::tuple<float, float, float, float> Camera::get_intrinsics() const {
        const float sx = float(_image_width) / float(_camera_width), sy = float(_image_height) / float(_camera_height);
        return {_focal_x * sx, _focal_y * sy, _center_x * sx, _center_y * sy};
    }
    torch::Tensor Camera::K() const {
        const auto [fx, fy, cx, cy] = get_intrinsics();
        return torch::tensor({fx, 0.f, cx, 0.f, fy, cy, 0.f, 0.f, 1.f}, torch::kFloat32).reshape({1, 3, 3}).to(_world_view_transform.device());
    }

    // src/core/splat_data.cpp:202-218
    SplatData::SplatData(int sh_degree, torch::Tensor means, torch::Tensor sh0, torch::Tensor shN, torch::Tensor scaling, torch::Tensor rotation,
                         torch::Tensor opacity, float scene_scale)
        : _max_sh_degree{sh_degree}, _active_sh_degree{0}, _scene_scale{scene_scale}, _means{std::move(means)}, _sh0{std::move(sh0)},
          _shN{std::move(shN)}, _scaling{std::move(scaling)}, _rotation{std::move(rotation)}, _opacity{std::move(opacity)} {}
    SplatData::~SplatData() {}   // (the reference waits for its asynchronous PLY saves here: none exist)
    // src/core/splat_data.cpp:267-286: the activations
    torch::Tensor SplatData::get_means() const { return _means; }
    torch::Tensor SplatData::get_opacity() const { return torch::sigmoid(_opacity).squeeze(-1); }
    torch::Tensor SplatData::get_rotation() const {
        return torch::nn::functional::normalize(_rotation, torch::nn::functional::NormalizeFuncOptions().dim(-1));
    }
    torch::Tensor SplatData::get_scaling() const { return torch::exp(_scaling); }
    torch::Tensor SplatData::get_shs() const { return torch::cat({_sh0, _shN}, 1); }
    void SplatData::set_active_sh_degree(int sh_degree) { _active_sh_degree = sh_degree <= _max_sh_degree ? sh_degree : _max_sh_degree; }   // src/core/splat_data.cpp:393-399

    namespace geometry {
        // only reached with a crop box, which the parity harness never passes: identity
        glm::mat4 EuclideanTransform::toMat4() const { return glm::mat4(1.0f); }
    }  // namespace geometry
}  // namespace gs
