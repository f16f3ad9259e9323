// TEST INFRASTRUCTURE ONLY — never used by the product.
//
// A driver of the reference's OWN PLY library (tinyply, vendored by the reference: include/external/tinyply.hpp + src/core/tinyply.cpp,
// compiled from where they lie by oracle/build_ref_ply.sh into oracle/_ref/ply_ref_tool) that writes a splat PLY through the same calls
// as the reference's exporter and parses one back:
//
//   ply_ref_tool write <blocks.bin> <out.ply>     blocks.bin = int32 n_rows, int32 n_blocks, then per block: int32 cols, int32 n_names,
//                                                 names ('\n'-joined, int32 length + bytes), float32 data [n_rows, cols] row-major.
//                                                 One tinyply::PlyFile::add_properties_to_element("vertex", names, FLOAT32, rows, data, INVALID, 0)
//                                                 per block, then write(os, binary = true): src/core/splat_data.cpp:135-162 (write_output_ply),
//                                                 blocks in the order of :119-133 (means, normals, sh0, shN, opacity, scaling, rotation).
//   ply_ref_tool read <in.ply> <out.bin>          parse_header, request every float property of "vertex" one by one, read; out.bin =
//                                                 int32 n_rows, int32 n_props, names ('\n'-joined, int32 length + bytes), float32 [n_props, n_rows].
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "external/tinyply.hpp"

static int32_t rd32(std::istream& is) { int32_t v; is.read(reinterpret_cast<char*>(&v), 4); return v; }
static void wr32(std::ostream& os, int32_t v) { os.write(reinterpret_cast<const char*>(&v), 4); }

static std::vector<std::string> split_lines(const std::string& s) {
    std::vector<std::string> out;
    std::stringstream ss(s);
    std::string l;
    while (std::getline(ss, l, '\n')) out.push_back(l);
    return out;
}

int main(int argc, char** argv) {
    if (argc != 4) { std::printf("usage: ply_ref_tool write <blocks.bin> <out.ply> | read <in.ply> <out.bin>\n"); return 1; }
    const std::string mode = argv[1];
    try {
        if (mode == "write") {
            std::ifstream in(argv[2], std::ios::binary);
            if (!in) throw std::runtime_error("cannot open blocks file");
            const int32_t rows = rd32(in), n_blocks = rd32(in);
            std::vector<std::vector<float>> data(n_blocks);
            tinyply::PlyFile ply;
            for (int b = 0; b < n_blocks; ++b) {
                const int32_t cols = rd32(in), n_names = rd32(in), len = rd32(in);
                std::string joined(len, '\0');
                in.read(joined.data(), len);
                std::vector<std::string> names = split_lines(joined);
                if ((int32_t)names.size() != n_names || n_names != cols) throw std::runtime_error("block names do not match its columns");
                data[b].resize((size_t)rows * cols);
                in.read(reinterpret_cast<char*>(data[b].data()), (std::streamsize)data[b].size() * 4);
                ply.add_properties_to_element("vertex", names, tinyply::Type::FLOAT32, (size_t)rows, reinterpret_cast<uint8_t*>(data[b].data()),
                                              tinyply::Type::INVALID, 0);
            }
            std::filebuf fb;
            fb.open(argv[3], std::ios::out | std::ios::binary);
            std::ostream os(&fb);
            ply.write(os, /*binary=*/true);
            return 0;
        }
        if (mode == "read") {
            std::ifstream in(argv[2], std::ios::binary);
            if (!in) throw std::runtime_error("cannot open ply");
            tinyply::PlyFile ply;
            ply.parse_header(in);
            std::vector<std::string> names;
            size_t rows = 0;
            for (const auto& e : ply.get_elements())
                if (e.name == "vertex") {
                    rows = e.size;
                    for (const auto& p : e.properties) names.push_back(p.name);
                }
            std::vector<std::shared_ptr<tinyply::PlyData>> cols;
            for (const auto& n : names) cols.push_back(ply.request_properties_from_element("vertex", {n}));
            ply.read(in);
            std::ofstream out(argv[3], std::ios::binary);
            wr32(out, (int32_t)rows);
            wr32(out, (int32_t)names.size());
            std::string joined;
            for (size_t i = 0; i < names.size(); ++i) joined += (i ? "\n" : "") + names[i];
            wr32(out, (int32_t)joined.size());
            out.write(joined.data(), (std::streamsize)joined.size());
            for (const auto& c : cols) {
                if (c->t != tinyply::Type::FLOAT32 || c->count != rows) throw std::runtime_error("non-float32 vertex property");
                out.write(reinterpret_cast<const char*>(c->buffer.get()), (std::streamsize)rows * 4);
            }
            return 0;
        }
    } catch (const std::exception& e) {
        std::fprintf(stderr, "ply_ref_tool: %s\n", e.what());
        return 2;
    }
    std::printf("unknown mode\n");
    return 1;
}
