// TEST INFRASTRUCTURE ONLY — stand-in for the reference's include/core/image_io.hpp (OpenImageIO): the COLMAP reader only asks for the
// size of the first image (get_image_info, src/loader/formats/colmap.cpp:856); colmap_ref_tool.cpp answers from the PNG header.
#pragma once
#include <filesystem>
#include <tuple>

std::tuple<int, int, int> get_image_info(std::filesystem::path p);
