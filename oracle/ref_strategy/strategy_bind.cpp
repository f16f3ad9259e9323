// TEST INFRASTRUCTURE ONLY (oracle/build_ref_strategy.sh).  Python entry to the reference's OWN training host logic — gs::training::MCMC
// (src/training/strategies/mcmc.cpp:85-505), strategy_utils.cpp:20-129, optimizers/fused_adam.cpp:20-119, scheduler.cpp — compiled UNMODIFIED and
// linked with this repository's drop-in (compat/gsplat + libgsx_gsplat_backend.so: gsplat::relocation / add_noise and fast_gs::optimizer::adam_step_wrapper),
// so that tests/test_gpu_reference_strategy.py can drive it iteration by iteration next to gsx.strategy.MCMC / gsx.optim.FusedAdam and compare what
// no kernel test sees: which Gaussians are dead, sampled and relocated, the optimizer-state surgery, the shN freeze, growth to max_cap, the lr schedule.
// The optimizer lives in a private member of MCMC; this binding TU (not the reference's sources) reads it through `#define private public`.
#include <torch/extension.h>

#define private public
#include "strategies/mcmc.hpp"
#undef private
#include "optimizers/fused_adam.hpp"
#include "rasterization/rasterizer.hpp"

namespace {
using gs::training::FusedAdam;

class RefMCMC {
public:
    RefMCMC(int sh_degree, int active_sh_degree, torch::Tensor means, torch::Tensor sh0, torch::Tensor shN, torch::Tensor scaling_raw, torch::Tensor rotation_raw,
            torch::Tensor opacity_raw, float scene_scale, py::dict p) {
        gs::SplatData model(sh_degree, means, sh0, shN, scaling_raw, rotation_raw, opacity_raw, scene_scale);
        model.set_active_sh_degree(active_sh_degree);
        gs::param::OptimizationParameters o;   // the struct's defaults (include/core/parameters.hpp:15-93), overridden field by field
        auto num = [&](const char* k, auto& dst) { if (p.contains(k)) dst = p[k].cast<std::remove_reference_t<decltype(dst)>>(); };
        num("iterations", o.iterations); num("sh_degree_interval", o.sh_degree_interval); num("means_lr", o.means_lr); num("shs_lr", o.shs_lr);
        num("opacity_lr", o.opacity_lr); num("scaling_lr", o.scaling_lr); num("rotation_lr", o.rotation_lr); num("min_opacity", o.min_opacity);
        num("refine_every", o.refine_every); num("start_refine", o.start_refine); num("stop_refine", o.stop_refine); num("max_cap", o.max_cap);
        strategy_ = std::make_unique<gs::training::MCMC>(std::move(model));
        strategy_->initialize(o);
    }
    // the six parameter tensors, in the optimizer's group order: means, sh0, shN, scaling_raw, rotation_raw, opacity_raw
    std::vector<torch::Tensor> params() {
        auto& m = strategy_->get_model();
        return {m.means(), m.sh0(), m.shN(), m.scaling_raw(), m.rotation_raw(), m.opacity_raw()};
    }
    void set_grads(std::vector<torch::Tensor> grads) {
        auto ps = params();
        TORCH_CHECK(grads.size() == 6, "six gradients");
        for (size_t i = 0; i < 6; ++i) ps[i].mutable_grad() = grads[i];
    }
    bool has_grad(int i) { return params()[i].grad().defined(); }
    void post_backward(int iter) {
        gs::training::RenderOutput out;   // MCMC::post_backward does not read it
        strategy_->post_backward(iter, out);
    }
    void step(int iter) { strategy_->step(iter); }
    bool is_refining(int iter) { return strategy_->is_refining(iter); }
    int active_sh_degree() { return strategy_->get_model().get_active_sh_degree(); }
    int64_t size() { return strategy_->get_model().size(); }
    double lr(int group) { return static_cast<FusedAdam::Options&>(strategy_->_optimizer->param_groups()[group].options()).lr(); }
    // (exp_avg, exp_avg_sq, step_count) of a group; undefined tensors and -1 while the group has no state yet
    std::tuple<torch::Tensor, torch::Tensor, int64_t> state(int group) {
        auto& opt = *strategy_->_optimizer;
        auto& param = opt.param_groups()[group].params()[0];
        auto it = opt.state().find(param.unsafeGetTensorImpl());
        if (it == opt.state().end()) return {torch::Tensor(), torch::Tensor(), -1};
        auto& st = static_cast<FusedAdam::AdamParamState&>(*it->second);
        return {st.exp_avg, st.exp_avg_sq, st.step_count};
    }
    // the pieces of a refine event on their own (private in the reference: MCMC::relocate_gs / add_new_gs / inject_noise, mcmc.cpp:114-366)
    int relocate_gs() { return strategy_->relocate_gs(); }
    int add_new_gs() { return strategy_->add_new_gs(); }
    void inject_noise() { strategy_->inject_noise(); }

private:
    std::unique_ptr<gs::training::MCMC> strategy_;
};
}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    py::class_<RefMCMC>(m, "RefMCMC")
        .def(py::init<int, int, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, float, py::dict>())
        .def("params", &RefMCMC::params).def("set_grads", &RefMCMC::set_grads).def("has_grad", &RefMCMC::has_grad)
        .def("post_backward", &RefMCMC::post_backward).def("step", &RefMCMC::step).def("is_refining", &RefMCMC::is_refining)
        .def("active_sh_degree", &RefMCMC::active_sh_degree).def("size", &RefMCMC::size).def("lr", &RefMCMC::lr).def("state", &RefMCMC::state)
        .def("relocate_gs", &RefMCMC::relocate_gs).def("add_new_gs", &RefMCMC::add_new_gs).def("inject_noise", &RefMCMC::inject_noise);
}
