// TEST INFRASTRUCTURE ONLY: forwards to the builder-written minimal glm subset (glm_min.hpp).
#pragma once
#include <glm/glm_min.hpp>
