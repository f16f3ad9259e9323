// TEST INFRASTRUCTURE ONLY (oracle/_ref build): the CUDA header name, served by HIP's cooperative groups.
#pragma once
#include <hip/hip_cooperative_groups.h>
