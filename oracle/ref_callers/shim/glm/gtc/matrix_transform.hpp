// TEST INFRASTRUCTURE ONLY (oracle/build_ref_callers.sh): the reference's include/geometry/euclidean_transform.hpp includes this glm header and
// calls glm::eulerAngles in an inline getter nothing on the render path uses.  The minimal glm subset of oracle/ref_hip/shim has neither.
#pragma once
#include <glm/glm.hpp>
#include <glm/gtc/quaternion.hpp>
namespace glm {
    template <typename T, qualifier Q>
    vec<3, T, Q> eulerAngles(const qua<T, Q>& q);   // declared only: never called by the compiled call sites
}
