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
    // src/core/camera.cpp:15-23
    static torch::Tensor world_to_view(const torch::Tensor& R, const torch::Tensor& t) {
        torch::Tensor w2c = torch::eye(4, torch::TensorOptions().dtype(torch::kFloat32).device(R.device()));
        w2c.index_put_({Slice(0, 3), Slice(0, 3)}, R);
        w2c.index_put_({Slice(0, 3), 3}, t);
        return w2c.to(torch::TensorOptions().dtype(torch::kFloat32).device(torch::kCUDA)).unsqueeze(0).contiguous();
    }

    // src/core/camera.cpp:25-58
    Camera::Camera(const torch::Tensor& R, const torch::Tensor& T, float focal_x, float focal_y, float center_x, float center_y,
                   const torch::Tensor radial_distortion, const torch::Tensor tangential_distortion, gsplat::CameraModelType camera_model_type,
                   const std::string& image_name, const std::filesystem::path& image_path, int camera_width, int camera_height, int uid)
        : _uid(uid), _focal_x(focal_x), _focal_y(focal_y), _center_x(center_x), _center_y(center_y), _R(R), _T(T),
          _radial_distortion(radial_distortion), _tangential_distortion(tangential_distortion), _camera_model_type(camera_model_type),
          _image_name(image_name), _image_path(image_path), _camera_width(camera_width), _camera_height(camera_height),
          _image_width(camera_width), _image_height(camera_height), _world_view_transform{world_to_view(R, T)} {
        auto c2w = torch::inverse(_world_view_transform.squeeze());
        _cam_position = c2w.index({Slice(None, 3), 3}).contiguous().squeeze();
        _FoVx = focal2fov(_focal_x, _camera_width);
        _FoVy = focal2fov(_focal_y, _camera_height);
    }

    // src/core/camera.cpp:82-103
    torch::Tensor Camera::K() const {
        const auto K = torch::zeros({1, 3, 3}, _world_view_transform.options());
        auto [fx, fy, cx, cy] = get_intrinsics();
        K[0][0][0] = fx;
        K[0][1][1] = fy;
        K[0][0][2] = cx;
        K[0][1][2] = cy;
        K[0][2][2] = 1.0f;
        return K;
    }
    std::tuple<float, float, float, float> Camera::get_intrinsics() const {
        const float xs = float(_image_width) / float(_camera_width), ys = float(_image_height) / float(_camera_height);
        return std::make_tuple(_focal_x * xs, _focal_y * ys, _center_x * xs, _center_y * ys);
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
