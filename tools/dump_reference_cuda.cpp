// dump_reference_cuda.cpp — emits the tensors of one `--gut` render + backward through the `gsplat::` operators of WHATEVER backend
// it is linked against, for a scene exported by tools/export_scene.py ("when a CUDA box is available, dump reference
// tensors for the §8d seeds").
//
//   On a CUDA box, inside a build of the reference (see tools/dump_reference_cuda.md): compile this file with the reference's
//   `gsplat/` on the include path and link its `gsplat_backend` — the dump then holds the REFERENCE CUDA kernels' outputs and
//   pins UT projection / intersect_offset / blend forward / blend backward, which no upstream test pins.
//   Here (tests/test_reference_dump.py, -m gpu): compiled against compat/gsplat + libgsx_gsplat_backend.so to prove the tool and the
//   loader work end to end; that self-dump is compared with the CPU oracle like any other output of this backend.
//
// File format (tests/golden/ref_dump.py reads it): <dir>/<name>.bin = raw little-endian array, <dir>/manifest.txt = one line per
// tensor "name dtype ndim d0 d1 ..." (dtype in f32 i32 i64 u8).  Inputs and outputs share the format.
//
// The op sequence and constants are those of gs::training::rasterize (src/training/rasterization/rasterizer.cpp:176-181, 248-329):
// eps2d 0.3, near 0.01, far 1e4, radius_clip 0, tile 16, GLOBAL shutter, default UT parameters, colours = clamp_min(SH + 0.5, 0).
#include <torch/torch.h>

#include "Ops.h"

#include <cstdio>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

namespace {

struct Entry { std::string dtype; std::vector<int64_t> shape; };

std::map<std::string, Entry> read_manifest(const std::string& dir) {
    std::map<std::string, Entry> m;
    std::ifstream f(dir + "/manifest.txt");
    TORCH_CHECK(f.good(), "cannot open ", dir, "/manifest.txt");
    std::string line;
    while (std::getline(f, line)) {
        std::istringstream ss(line);
        std::string name;
        Entry e;
        int nd = 0;
        if (!(ss >> name >> e.dtype >> nd)) continue;
        for (int i = 0; i < nd; ++i) { int64_t d; ss >> d; e.shape.push_back(d); }
        m[name] = e;
    }
    return m;
}

torch::ScalarType to_scalar_type(const std::string& d) {
    if (d == "f32") return torch::kFloat32;
    if (d == "i32") return torch::kInt32;
    if (d == "i64") return torch::kInt64;
    if (d == "u8") return torch::kUInt8;
    TORCH_CHECK(false, "unknown dtype ", d);
}

const char* dtype_name(torch::ScalarType t) {
    switch (t) {
        case torch::kFloat32: return "f32";
        case torch::kInt32: return "i32";
        case torch::kInt64: return "i64";
        case torch::kUInt8: case torch::kBool: return "u8";
        default: TORCH_CHECK(false, "unsupported dtype in dump");
    }
}

torch::Tensor load(const std::string& dir, const std::map<std::string, Entry>& man, const std::string& name) {
    auto it = man.find(name);
    TORCH_CHECK(it != man.end(), "scene has no tensor '", name, "'");
    torch::Tensor t = torch::empty(it->second.shape, torch::TensorOptions().dtype(to_scalar_type(it->second.dtype)));
    std::ifstream f(dir + "/" + name + ".bin", std::ios::binary);
    TORCH_CHECK(f.good(), "cannot open ", dir, "/", name, ".bin");
    f.read(reinterpret_cast<char*>(t.data_ptr()), (std::streamsize)t.nbytes());
    TORCH_CHECK((size_t)f.gcount() == t.nbytes(), name, ".bin is shorter than its manifest entry");
    return t;
}

struct Dump {
    std::string dir;
    std::ofstream manifest;
    explicit Dump(const std::string& d) : dir(d), manifest(d + "/manifest.txt") { TORCH_CHECK(manifest.good(), "cannot write into ", d); }
    void put(const std::string& name, const torch::Tensor& t_) {
        torch::Tensor t = t_.detach().to(torch::kCPU).contiguous();
        if (t.scalar_type() == torch::kBool) t = t.to(torch::kUInt8);
        std::ofstream f(dir + "/" + name + ".bin", std::ios::binary);
        f.write(reinterpret_cast<const char*>(t.data_ptr()), (std::streamsize)t.nbytes());
        manifest << name << " " << dtype_name(t.scalar_type()) << " " << t.dim();
        for (auto d : t.sizes()) manifest << " " << d;
        manifest << "\n";
    }
};

}  // namespace

int main(int argc, char** argv) {
    if (argc < 3) {
        std::printf("usage: %s <scene_dir (tools/export_scene.py)> <out_dir (must exist)>\n", argv[0]);
        return 1;
    }
    try {
        const std::string in = argv[1], out = argv[2];
        const auto man = read_manifest(in);
        const torch::Device dev(torch::kCUDA, 0);
        auto L = [&](const char* n) { return load(in, man, n).to(dev).contiguous(); };
        const torch::Tensor means = L("means"), quats = L("quats"), scales = L("scales"), opacities = L("opacities"), sh = L("sh");
        const torch::Tensor viewmat = L("viewmat").reshape({1, 4, 4}), K = L("K").reshape({1, 3, 3});
        const torch::Tensor bg = L("background").reshape({1, 3});
        const torch::Tensor v_render_colors = L("v_render_colors"), v_render_alphas = L("v_render_alphas");
        const torch::Tensor dims = load(in, man, "dims");  // [width, height, sh_degree] int32
        const int width = dims[0].item<int>(), height = dims[1].item<int>(), sh_degree = dims[2].item<int>();
        const int tile_size = 16;
        const int64_t N = means.size(0);
        UnscentedTransformParameters ut_params;
        std::optional<torch::Tensor> none;
        Dump d(out);

        auto proj = gsplat::projection_ut_3dgs_fused(means, quats, scales, opacities, viewmat, std::nullopt, K, width, height, 0.3f, 0.01f, 10000.f, 0.f,
                                                     false, gsplat::CameraModelType::PINHOLE, ut_params, ShutterType::GLOBAL, none, none, none);
        const torch::Tensor radii = std::get<0>(proj).contiguous(), means2d = std::get<1>(proj).contiguous(), depths = std::get<2>(proj).contiguous();
        const torch::Tensor valid = (radii > 0).all(-1);
        d.put("radii", radii);
        // culled rows of means2d / depths / conics are not written by the kernel (ProjectionUT3DGSFused.cu:78-82): zero them so dumps compare
        d.put("means2d", means2d * valid.unsqueeze(-1));
        d.put("depths", depths * valid);
        d.put("conics", std::get<3>(proj) * valid.unsqueeze(-1));

        const torch::Tensor campos = torch::inverse(viewmat).index({torch::indexing::Slice(), torch::indexing::Slice(0, 3), 3});
        const torch::Tensor dirs = (means.unsqueeze(0) - campos.unsqueeze(1)).reshape({-1, 3}).contiguous();
        const torch::Tensor coeffs = sh.reshape({N, -1, 3}).contiguous();
        torch::Tensor colors = gsplat::spherical_harmonics_fwd(sh_degree, dirs, coeffs, valid.reshape({-1}));
        colors = (torch::clamp_min(colors + 0.5f, 0.f) * valid.reshape({-1, 1})).reshape({1, N, 3}).contiguous();
        d.put("colors", colors);

        const int tile_width = (width + tile_size - 1) / tile_size, tile_height = (height + tile_size - 1) / tile_size;
        const auto isect = gsplat::intersect_tile(means2d, radii, depths, {}, {}, 1, tile_size, tile_width, tile_height, true);
        const torch::Tensor isect_ids = std::get<1>(isect), flatten_ids = std::get<2>(isect);
        const torch::Tensor isect_offsets = gsplat::intersect_offset(isect_ids, 1, tile_width, tile_height).reshape({1, tile_height, tile_width});
        d.put("tiles_per_gauss", std::get<0>(isect));
        d.put("isect_ids", isect_ids);
        d.put("flatten_ids", flatten_ids);
        d.put("isect_offsets", isect_offsets);

        const torch::Tensor opac = opacities.unsqueeze(0).contiguous();
        auto fwd = gsplat::rasterize_to_pixels_from_world_3dgs_fwd(means, quats, scales, colors, opac, bg, std::nullopt, width, height, tile_size, viewmat,
                                                                   std::nullopt, K, gsplat::CameraModelType::PINHOLE, ut_params, ShutterType::GLOBAL, none,
                                                                   none, none, isect_offsets.contiguous(), flatten_ids.contiguous());
        d.put("renders", std::get<0>(fwd));
        d.put("alphas", std::get<1>(fwd));
        d.put("last_ids", std::get<2>(fwd));
        auto bwd = gsplat::rasterize_to_pixels_from_world_3dgs_bwd(means, quats, scales, colors, opac, bg, std::nullopt, width, height, tile_size, viewmat,
                                                                   std::nullopt, K, gsplat::CameraModelType::PINHOLE, ut_params, ShutterType::GLOBAL, none,
                                                                   none, none, isect_offsets.contiguous(), flatten_ids.contiguous(), std::get<1>(fwd),
                                                                   std::get<2>(fwd), v_render_colors, v_render_alphas);
        d.put("v_means", std::get<0>(bwd));
        d.put("v_quats", std::get<1>(bwd));
        d.put("v_scales", std::get<2>(bwd));
        d.put("v_colors", std::get<3>(bwd));
        d.put("v_opacities", std::get<4>(bwd));
        auto shg = gsplat::spherical_harmonics_bwd(coeffs.size(1), sh_degree, dirs, coeffs, valid.reshape({-1}), std::get<3>(bwd).reshape({-1, 3}).contiguous(), true);
        d.put("v_sh", std::get<0>(shg));
        d.put("v_dirs", std::get<1>(shg));
        torch::cuda::synchronize();
        std::printf("dumped %lld Gaussians, %lld intersections, %dx%d into %s\n", (long long)N, (long long)flatten_ids.size(0), width, height, out.c_str());
        return 0;
    } catch (const std::exception& e) {
        std::printf("EXCEPTION: %s\n", e.what());
        return 2;
    }
}
