// TEST INFRASTRUCTURE ONLY (oracle/_ref build): cub::DeviceRadixSort / cub::DoubleBuffer served by hipCUB.
#pragma once
#include <hipcub/hipcub.hpp>
namespace cub = hipcub;
