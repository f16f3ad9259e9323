// TEST INFRASTRUCTURE ONLY (oracle/build_ref_strategy.sh).  The three more members of the reference's gs::SplatData that its training host logic
// (src/training/strategies/mcmc.cpp, strategy_utils.cpp — compiled UNMODIFIED) references on top of what oracle/ref_callers/core_standins.cpp
// defines for the render call sites: the move operations and increment_sh_degree, restated against the reference's own header
// (include/core/splat_data.hpp, read where it lies) from src/core/splat_data.cpp:221-260,387-391 — which lives next to PLY / SOG export
// (tinyply, TBB, OpenImageIO: not in this image).  Never linked into the product.
#include "core/splat_data.hpp"

namespace gs {
    // src/core/splat_data.cpp:221-235 (the save mutex / futures are default constructed)
    SplatData::SplatData(SplatData&& other) noexcept
        : _densification_info(std::move(other._densification_info)), _active_sh_degree(other._active_sh_degree), _max_sh_degree(other._max_sh_degree),
          _scene_scale(other._scene_scale), _means(std::move(other._means)), _sh0(std::move(other._sh0)), _shN(std::move(other._shN)),
          _scaling(std::move(other._scaling)), _rotation(std::move(other._rotation)), _opacity(std::move(other._opacity)) {}

    // src/core/splat_data.cpp:238-260 (no asynchronous saves exist here: nothing to wait for)
    SplatData& SplatData::operator=(SplatData&& other) noexcept {
        if (this != &other) {
            _active_sh_degree = other._active_sh_degree;
            _max_sh_degree = other._max_sh_degree;
            _scene_scale = other._scene_scale;
            _means = std::move(other._means);
            _sh0 = std::move(other._sh0);
            _shN = std::move(other._shN);
            _scaling = std::move(other._scaling);
            _rotation = std::move(other._rotation);
            _opacity = std::move(other._opacity);
            _densification_info = std::move(other._densification_info);
        }
        return *this;
    }

    // src/core/splat_data.cpp:387-391
    void SplatData::increment_sh_degree() {
        if (_active_sh_degree < _max_sh_degree) _active_sh_degree++;
    }
}  // namespace gs
