// TEST INFRASTRUCTURE ONLY (oracle/build_ref_strategy.sh).  The reference's include/core/parameters.hpp names nlohmann::json in the DECLARATIONS of
// OptimizationParameters::to_json / from_json, which the compiled strategy sources never call; nlohmann/json is a vcpkg dependency that is not in this
// image.  A declaration is all a function declaration needs.
#pragma once
#include <optional>   // (the real <expected> / json_fwd.hpp bring it in transitively; parameters.hpp:112 relies on that)
namespace nlohmann {
    class json;
}
