// TEST INFRASTRUCTURE ONLY — driver around the reference's own COLMAP reader (src/loader/formats/colmap.cpp, compiled unmodified from
// where it lies by oracle/build_ref_colmap.sh).  Reads a COLMAP model with gs::loader::read_colmap_cameras_and_images(_text) and
// read_colmap_point_cloud(_text) and prints what the reference parsed, one record per line, floats with 9 significant digits
// (round-trip exact for fp32):
//   center cx cy cz
//   camera uid camera_id model model_type width height fx fy cx cy | name | R(9) | T(3) | radial.. | tangential..
//   points N          followed by N lines "x y z r g b"
// Usage: colmap_ref_tool <base> <images_folder> bin|text
#include "colmap.hpp"
#include "core/image_io.hpp"

#include <cstdio>
#include <fstream>
#include <string>

// PNG only (IHDR): width / height big-endian at bytes 16..23, colour type at 25
std::tuple<int, int, int> get_image_info(std::filesystem::path p) {
    std::ifstream f(p, std::ios::binary);
    unsigned char h[26] = {0};
    f.read(reinterpret_cast<char*>(h), 26);
    if (!f || h[1] != 'P' || h[2] != 'N' || h[3] != 'G') throw std::runtime_error("get_image_info: not a PNG: " + p.string());
    const int w = (h[16] << 24) | (h[17] << 16) | (h[18] << 8) | h[19], ht = (h[20] << 24) | (h[21] << 16) | (h[22] << 8) | h[23];
    const int ch = h[25] == 6 ? 4 : h[25] == 2 ? 3 : h[25] == 4 ? 2 : 1;
    return {w, ht, ch};
}

static void print_tensor(const torch::Tensor& t) {
    auto c = t.to(torch::kCPU).to(torch::kFloat32).contiguous().flatten();
    for (int64_t i = 0; i < c.numel(); ++i) std::printf(" %.9g", c[i].item<float>());
}

int main(int argc, char** argv) {
    if (argc != 4) { std::fprintf(stderr, "usage: colmap_ref_tool <base> <images_folder> bin|text\n"); return 2; }
    const std::filesystem::path base = argv[1];
    const std::string folder = argv[2], mode = argv[3];
    try {
        auto [cams, center] = mode == "text" ? gs::loader::read_colmap_cameras_and_images_text(base, folder)
                                             : gs::loader::read_colmap_cameras_and_images(base, folder);
        std::printf("center");
        print_tensor(center);
        std::printf("\n");
        for (size_t i = 0; i < cams.size(); ++i) {
            const auto& c = cams[i];
            std::printf("camera %zu %u %d %d %d %d %.9g %.9g %.9g %.9g | %s |", i, c._camera_ID, (int)c._camera_model, (int)c._camera_model_type, c._width,
                        c._height, c._focal_x, c._focal_y, c._center_x, c._center_y, c._image_name.c_str());
            print_tensor(c._R);
            std::printf(" |");
            print_tensor(c._T);
            std::printf(" |");
            print_tensor(c._radial_distortion);
            std::printf(" |");
            print_tensor(c._tangential_distortion);
            std::printf("\n");
        }
        auto pc = mode == "text" ? gs::loader::read_colmap_point_cloud_text(base) : gs::loader::read_colmap_point_cloud(base);
        auto m = pc.means.to(torch::kCPU).to(torch::kFloat32).contiguous();
        auto col = pc.colors.to(torch::kCPU).to(torch::kFloat32).contiguous();
        std::printf("points %lld\n", (long long)pc.size());
        for (int64_t i = 0; i < pc.size(); ++i)
            std::printf("%.9g %.9g %.9g %.9g %.9g %.9g\n", m[i][0].item<float>(), m[i][1].item<float>(), m[i][2].item<float>(), col[i][0].item<float>(),
                        col[i][1].item<float>(), col[i][2].item<float>());
    } catch (const std::exception& e) {
        std::printf("error %s\n", e.what());
        return 1;
    }
    return 0;
}
