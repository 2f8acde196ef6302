// molann_torch.cpp - the plans of libmolann_hip.so as TorchScript operators.
//
// The reference's own tests end every case with torch.jit.script(module).save(...) (test/test_molann.py:36,
// 46,62,75,101,114) and README.rst:49 ships models to MD engines that way.  A scripted module cannot call
// ctypes, so the same C ABI (include/molann_hip.h) is bound here a second time, as dispatcher operators a
// TorchScript graph can name:
//
//     molann::run(Tensor x, int[] desc, Tensor ref_x, Tensor[] weights, Tensor[] biases) -> Tensor
//     molann::run_backward(Tensor x, int[] desc, Tensor ref_x, Tensor[] weights, Tensor[] biases,
//                          Tensor grad_out, bool need_x, bool need_params) -> Tensor[]
//
// `desc` is everything the modules fix at construction time, as integers (layout below; written by
// molann_amd/script.py).  Plans are created on first use and cached per (desc, device); desc carries an
// instance id, so two models of the same architecture have plans (packed weights, workspaces) of their own.
// The live tensors (ref_x buffer, Linear parameters) are re-read whenever one of them is a different tensor
// OBJECT than the one packed last (identity through a weak reference to its TensorImpl: an address the
// allocator hands out again after a model was freed cannot pass for the old tensor), or its storage or version
// counter changed.  The one edit this cannot see is an in-place write through `.data` / under a detached alias
// (it bumps the alias's counter): molann::invalidate(desc, device) - `model.refresh_parameters()` in Python -
// or MOLANN_ALWAYS_REPACK=1 covers that.  Entries are evicted least-recently-used beyond
// MOLANN_PLAN_CACHE_SIZE (default 64) and by molann::release / molann::drop_plans.
// Only the HIP dispatch key is registered: a CPU tensor raises, there is no fallback.
// A libtorch host loads this library (dlopen / torch.ops.load_library) before torch::jit::load.
//
// desc: [0]=2 (layout version) [1]=kind (0 align, 1 features, 2 features+MLP) [2]=n_inp [3]=n_align
//       [4]=n_features [5]=use_angle_value [6]=n_layers [7]=activation [8]=mlp_precision [9]=instance id, then
//       align_idx[n_align], feat_type[n_features], feat_ptr[n_features+1], feat_idx[feat_ptr[n_features]],
//       layer_dims[n_layers+1] (only when n_layers > 0)

#include <torch/library.h>
#include <torch/csrc/autograd/custom_function.h>
#include <torch/csrc/autograd/autograd.h>
#include <ATen/ATen.h>
#include <c10/core/DeviceGuard.h>
#include <c10/hip/HIPStream.h>

#include <cstdlib>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "molann_hip.h"

namespace {

enum { KIND_ALIGN = 0, KIND_FEATURES = 1, KIND_FORWARD = 2, DESC_LAYOUT = 2, DESC_HEAD = 10 };

// What was packed: the tensor OBJECT (a weak reference keeps the TensorImpl's address from being reused while we
// remember it, and says when the tensor is gone), its storage address and its version counter.
struct TensorKey {
    c10::weak_intrusive_ptr<c10::TensorImpl> impl{c10::weak_intrusive_ptr<c10::TensorImpl>(
        c10::intrusive_ptr<c10::TensorImpl, c10::UndefinedTensorImpl>())};
    const void* ptr = nullptr;
    int64_t version = -1;
    bool matches(const at::Tensor& t) const {
        return version >= 0 && !impl.expired() && impl._unsafe_get_target() == t.unsafeGetTensorImpl() && ptr == t.data_ptr() &&
               version == (int64_t)t._version();
    }
};

TensorKey key_of(const at::Tensor& t) {
    TensorKey k;
    k.impl = c10::weak_intrusive_ptr<c10::TensorImpl>(t.getIntrusivePtr());
    k.ptr = t.data_ptr();
    k.version = (int64_t)t._version();
    return k;
}

struct Entry {
    molann_plan* plan = nullptr;
    int kind = 0, n_inp = 0, n_align = 0, n_layers = 0, out_dim = 0, feature_dim = 0;
    TensorKey ref_key;
    std::vector<TensorKey> mlp_key;
    bool dirty = false; // molann::invalidate: repack at the next call whatever the keys say
    uint64_t last_use = 0;
    int pins = 0;  // captured HIP graphs that launch through this plan (molann::pin): never evicted while > 0 (guarded by g_cache_mu)
    std::mutex mu; // update_* + launch of one plan are one critical section
    ~Entry() {
        if (plan) molann_plan_destroy(plan);
    }
};

void check(int rc, const char* what) {
    TORCH_CHECK(rc == 0, what, " failed: ", molann_error_string(rc), " (", rc, ")");
}

struct Parsed {
    std::vector<int32_t> align_idx, feat_type, feat_ptr, feat_idx, layer_dims;
    molann_plan_desc d;
};

// the integer list -> molann_plan_desc (pointers into `p`)
void parse_desc(const std::vector<int64_t>& v, Parsed& p) {
    TORCH_CHECK(v.size() >= DESC_HEAD && v[0] == DESC_LAYOUT, "molann::run: unknown descriptor layout");
    const int64_t n_align = v[3], n_feat = v[4], n_layers = v[6];
    TORCH_CHECK(n_align >= 0 && n_feat >= 0 && n_layers >= 0 && n_layers <= MOLANN_MAX_LAYERS, "molann::run: bad descriptor counts");
    size_t pos = DESC_HEAD;
    auto take = [&](std::vector<int32_t>& out, int64_t n) {
        TORCH_CHECK(pos + (size_t)n <= v.size(), "molann::run: descriptor too short");
        out.assign(v.begin() + pos, v.begin() + pos + n);
        pos += (size_t)n;
    };
    take(p.align_idx, n_align);
    take(p.feat_type, n_feat);
    take(p.feat_ptr, n_feat > 0 ? n_feat + 1 : 0);
    take(p.feat_idx, n_feat > 0 ? p.feat_ptr.back() : 0);
    take(p.layer_dims, n_layers > 0 ? n_layers + 1 : 0);
    TORCH_CHECK(pos == v.size(), "molann::run: descriptor has trailing entries");
    molann_plan_desc& d = p.d;
    d = molann_plan_desc();
    d.abi_version = MOLANN_ABI_VERSION;
    d.n_inp = (int32_t)v[2];
    d.n_align = (int32_t)n_align;
    d.align_idx = p.align_idx.data();
    d.n_features = (int32_t)n_feat;
    d.feat_type = p.feat_type.data();
    d.feat_ptr = p.feat_ptr.data();
    d.feat_idx = p.feat_idx.data();
    d.use_angle_value = (int32_t)v[5];
    d.n_layers = (int32_t)n_layers;
    d.layer_dims = p.layer_dims.data();
    d.activation = (int32_t)v[7];
    d.mlp_precision = (int32_t)v[8];
}

typedef std::pair<std::vector<int64_t>, int> CacheKey;
std::mutex g_cache_mu;
std::map<CacheKey, std::shared_ptr<Entry>> g_cache;
uint64_t g_clock = 0;

size_t cache_capacity() {
    static const size_t cap = [] {
        const char* e = getenv("MOLANN_PLAN_CACHE_SIZE");
        const long v = e ? atol(e) : 64;
        return (size_t)(v < 1 ? 1 : v);
    }();
    return cap;
}

bool always_repack() {
    static const bool v = [] { const char* e = getenv("MOLANN_ALWAYS_REPACK"); return e && e[0] == '1'; }();
    return v;
}

// caller holds g_cache_mu.  An evicted entry that a running call still holds lives until that call returns
// (shared_ptr); its plan and device memory go with the last reference.
// Entries pinned by a captured graph are not candidates: a graph replay never passes through entry_for (its
// last_use does not move) and holds raw pointers into the plan's device memory and code objects.
void evict_lru_locked() {
    while (g_cache.size() > cache_capacity()) {
        auto victim = g_cache.end();
        for (auto it = g_cache.begin(); it != g_cache.end(); ++it)
            if (it->second->pins == 0 && (victim == g_cache.end() || it->second->last_use < victim->second->last_use)) victim = it;
        if (victim == g_cache.end()) return;   // everything left is pinned
        g_cache.erase(victim);
    }
}

// molann_plan_create reads two switches from the environment (MOLANN_NO_JIT, MOLANN_NO_REGS: which kernel
// family serves the plan); a plan built under other settings must not be handed out, so they are part of the key
int cache_device_key(int device) {
    const char* a = getenv("MOLANN_NO_JIT");
    const char* b = getenv("MOLANN_NO_REGS");
    return device | ((a && a[0] == '1') ? 1 << 16 : 0) | ((b && b[0] == '1') ? 1 << 17 : 0);
}

// plan for (desc, device of x); created with the current contents of ref_x
std::shared_ptr<Entry> entry_for(const std::vector<int64_t>& desc, const at::Tensor& x, const at::Tensor& ref_x) {
    const auto key = std::make_pair(desc, cache_device_key((int)x.get_device()));
    std::lock_guard<std::mutex> lock(g_cache_mu);
    auto it = g_cache.find(key);
    if (it != g_cache.end()) {
        it->second->last_use = ++g_clock;
        return it->second;
    }
    Parsed p;
    parse_desc(desc, p);
    at::Tensor ref_host;
    if (p.d.n_align > 0) {
        TORCH_CHECK(ref_x.numel() == 3 * (int64_t)p.d.n_align, "molann::run: ref_x must be [n_align, 3]");
        ref_host = ref_x.detach().to(at::kCPU, at::kFloat).contiguous();
        p.d.ref_x = ref_host.data_ptr<float>();
    }
    auto e = std::make_shared<Entry>();
    check(molann_plan_create(&p.d, &e->plan), "molann_plan_create");
    e->kind = (int)desc[1];
    e->n_inp = p.d.n_inp;
    e->n_align = p.d.n_align;
    e->n_layers = p.d.n_layers;
    e->out_dim = molann_plan_out_dim(e->plan);
    e->feature_dim = molann_plan_feature_dim(e->plan);
    e->last_use = ++g_clock;
    g_cache.emplace(key, e);
    evict_lru_locked();
    return e;
}

// AlignmentLayer on its own has no backward kernel; the same map written as alignment + one position item
// per atom does (molann_amd/ann.py does the same)
std::vector<int64_t> align_as_features(const std::vector<int64_t>& desc) {
    const int64_t n_inp = desc[2], n_align = desc[3];
    std::vector<int64_t> v(desc.begin(), desc.begin() + DESC_HEAD + n_align);
    v[1] = KIND_FEATURES;
    v[4] = 1; // one position feature over all atoms
    v.push_back(MOLANN_FEAT_POSITION);
    v.push_back(0);
    v.push_back(n_inp);
    for (int64_t i = 0; i < n_inp; ++i) v.push_back(i);
    return v;
}

at::Tensor device_f32(const at::Tensor& t, const at::Tensor& like, const char* name) {
    TORCH_CHECK(t.scalar_type() == at::kFloat, "molann::run: ", name, " must be float32, got ", t.scalar_type());
    TORCH_CHECK(t.device() == like.device(), "molann::run: ", name, " is on ", t.device(), " but x is on ", like.device());
    return t.contiguous();
}

// bring the plan's copies of the live tensors up to date (caller holds e.mu, device guard set)
void sync_live(Entry& e, const at::Tensor& x, const at::Tensor& ref_x, const std::vector<at::Tensor>& weights,
               const std::vector<at::Tensor>& biases, hipStream_t stream) {
    if (e.n_align > 0) {
        TORCH_CHECK(ref_x.scalar_type() == x.scalar_type(), "molann::run: expected ref_x and x to have the same dtype, but got ",
                    ref_x.scalar_type(), " and ", x.scalar_type());
        TORCH_CHECK(ref_x.device() == x.device(), "molann::run: ref_x is on ", ref_x.device(), " but x is on ", x.device());
        if (e.dirty || always_repack() || !e.ref_key.matches(ref_x)) {
            const at::Tensor r = ref_x.detach().contiguous();
            if (r.scalar_type() == at::kDouble) check(molann_plan_update_ref_f64(e.plan, r.data_ptr<double>(), stream), "molann_plan_update_ref_f64");
            else check(molann_plan_update_ref(e.plan, r.data_ptr<float>(), stream), "molann_plan_update_ref");
            e.ref_key = key_of(ref_x);
        }
    }
    if (x.scalar_type() == at::kDouble) { e.dirty = false; return; }   // float64: the Linear parameters are read as they are
    if (e.kind == KIND_FORWARD) {
        TORCH_CHECK((int)weights.size() == e.n_layers && (int)biases.size() == e.n_layers,
                    "molann::run: expected ", e.n_layers, " weight and bias tensors");
        bool changed = e.dirty || always_repack() || e.mlp_key.size() != 2 * (size_t)e.n_layers;
        for (int l = 0; !changed && l < e.n_layers; ++l)
            changed = !e.mlp_key[2 * l].matches(weights[l]) || !e.mlp_key[2 * l + 1].matches(biases[l]);
        if (changed) {
            std::vector<at::Tensor> hold;
            std::vector<const float*> W, B;
            for (int l = 0; l < e.n_layers; ++l) {
                hold.push_back(device_f32(weights[l].detach(), x, "weight"));
                W.push_back(hold.back().data_ptr<float>());
                hold.push_back(device_f32(biases[l].detach(), x, "bias"));
                B.push_back(hold.back().data_ptr<float>());
            }
            check(molann_plan_update_mlp(e.plan, W.data(), B.data(), stream), "molann_plan_update_mlp");
            e.mlp_key.clear();
            for (int l = 0; l < e.n_layers; ++l) { e.mlp_key.push_back(key_of(weights[l])); e.mlp_key.push_back(key_of(biases[l])); }
        }
    }
    e.dirty = false;
}

void check_x(const at::Tensor& x, const std::vector<int64_t>& desc) {
    TORCH_CHECK(desc.size() >= DESC_HEAD, "molann::run: bad descriptor");
    TORCH_CHECK(x.dim() == 3 && x.size(1) == desc[2] && x.size(2) == 3, "Input should be a 3d torch tensor, with sizes [*, ",
                desc[2], ", 3]. Actual sizes: ", x.sizes());
    TORCH_CHECK(x.scalar_type() == at::kFloat || x.scalar_type() == at::kDouble, "molann_amd kernels are float32 / float64; got ",
                x.scalar_type());
}

at::Tensor run_impl(const at::Tensor& x_in, const std::vector<int64_t>& desc, const at::Tensor& ref_x, const std::vector<at::Tensor>& weights,
                    const std::vector<at::Tensor>& biases);
// A description registered once (molann::register_desc) and named by a small integer afterwards: what the eager modules'
// inference calls use - converting the 30-odd integers of a description from a Python list at every call costs more
// than the launch's own bookkeeping.  Scripted modules keep the self-contained list form (molann::run).
std::mutex g_handle_mu;
std::vector<std::shared_ptr<const std::vector<int64_t>>> g_handles;

int64_t register_desc(std::vector<int64_t> desc) {
    TORCH_CHECK(desc.size() >= DESC_HEAD && desc[0] == DESC_LAYOUT, "molann::register_desc: unknown descriptor layout");
    std::lock_guard<std::mutex> lock(g_handle_mu);
    for (size_t i = 0; i < g_handles.size(); ++i)
        if (*g_handles[i] == desc) return (int64_t)i;
    g_handles.push_back(std::make_shared<const std::vector<int64_t>>(std::move(desc)));
    return (int64_t)g_handles.size() - 1;
}

at::Tensor run_h_hip(const at::Tensor& x, int64_t handle, const at::Tensor& ref_x, std::vector<at::Tensor> weights, std::vector<at::Tensor> biases) {
    std::shared_ptr<const std::vector<int64_t>> d;
    {
        std::lock_guard<std::mutex> lock(g_handle_mu);
        TORCH_CHECK(handle >= 0 && (size_t)handle < g_handles.size(), "molann::run_h: unknown handle ", handle);
        d = g_handles[(size_t)handle];
    }
    return run_impl(x, *d, ref_x, weights, biases);
}

at::Tensor run_hip(const at::Tensor& x_in, std::vector<int64_t> desc, const at::Tensor& ref_x, std::vector<at::Tensor> weights,
                   std::vector<at::Tensor> biases) {
    return run_impl(x_in, desc, ref_x, weights, biases);
}
at::Tensor run_impl(const at::Tensor& x_in, const std::vector<int64_t>& desc, const at::Tensor& ref_x, const std::vector<at::Tensor>& weights,
                    const std::vector<at::Tensor>& biases) {
    check_x(x_in, desc);
    const at::Tensor x = x_in.contiguous();
    const c10::DeviceGuard guard(x.device());
    auto e = entry_for(desc, x, ref_x);
    const int64_t n = x.size(0);
    at::Tensor out = e->kind == KIND_ALIGN ? at::empty_like(x)
                                           : at::empty({n, e->kind == KIND_FORWARD ? e->out_dim : e->feature_dim}, x.options());
    if (n == 0) return out;
    hipStream_t stream = c10::hip::getCurrentHIPStream(x.get_device()).stream();
    std::lock_guard<std::mutex> lock(e->mu);
    sync_live(*e, x, ref_x, weights, biases, stream);
    if (x.scalar_type() == at::kDouble) { // `model.double()(x.double())`: the float64 entry points
        const double* xd = x.data_ptr<double>();
        double* od = out.data_ptr<double>();
        if (e->kind == KIND_ALIGN) check(molann_align_f64(e->plan, xd, n, od, stream), "molann_align_f64");
        else if (e->kind == KIND_FEATURES) check(molann_features_f64(e->plan, xd, n, od, stream), "molann_features_f64");
        else {
            TORCH_CHECK((int)weights.size() == e->n_layers && (int)biases.size() == e->n_layers, "molann::run: expected ", e->n_layers,
                        " weight and bias tensors");
            std::vector<at::Tensor> hold;
            std::vector<const double*> W, B;
            for (int l = 0; l < e->n_layers; ++l) {
                TORCH_CHECK(weights[l].scalar_type() == at::kDouble && biases[l].scalar_type() == at::kDouble && weights[l].device() == x.device(),
                            "molann::run: ann_layers must be float64 on ", x.device(), " for a float64 input");
                hold.push_back(weights[l].detach().contiguous()); W.push_back(hold.back().data_ptr<double>());
                hold.push_back(biases[l].detach().contiguous()); B.push_back(hold.back().data_ptr<double>());
            }
            at::Tensor work = at::empty({n, e->feature_dim}, x.options());
            check(molann_forward_f64(e->plan, xd, n, W.data(), B.data(), work.data_ptr<double>(), od, stream), "molann_forward_f64");
        }
        return out;
    }
    const float* xp = x.data_ptr<float>();
    float* op = out.data_ptr<float>();
    if (e->kind == KIND_ALIGN) check(molann_align_f32(e->plan, xp, n, op, stream), "molann_align_f32");
    else if (e->kind == KIND_FEATURES) check(molann_features_f32(e->plan, xp, n, op, stream), "molann_features_f32");
    else check(molann_forward_packed_f32(e->plan, xp, n, op, stream), "molann_forward_packed_f32");
    return out;
}

// [grad_x or empty, flat parameter gradients (dW_l, db_l per layer) or empty]
std::vector<at::Tensor> run_backward_hip(const at::Tensor& x_in, std::vector<int64_t> desc, const at::Tensor& ref_x,
                                         std::vector<at::Tensor> weights, std::vector<at::Tensor> biases,
                                         const at::Tensor& grad_out, bool need_x, bool need_params) {
    check_x(x_in, desc);
    const at::Tensor x = x_in.contiguous();
    const c10::DeviceGuard guard(x.device());
    const bool align_only = desc[1] == KIND_ALIGN;
    auto e = entry_for(align_only ? align_as_features(desc) : desc, x, ref_x);
    TORCH_CHECK(x.scalar_type() == at::kDouble || molann_plan_supports_backward(e->plan) == 1,
                "no backward kernel for this plan (wide MLP / large frames / this activation): run it under torch.no_grad()");
    const int64_t n = x.size(0);
    const int64_t cols = e->kind == KIND_FORWARD ? e->out_dim : e->feature_dim;
    if (x.scalar_type() == at::kDouble) { // float64: the features' backward in double (the MLP of a float64 model is ATen's)
        TORCH_CHECK(e->kind != KIND_FORWARD && !need_params, "molann::run_backward: float64 gradients exist for the preprocessing only");
        at::Tensor g = grad_out.to(at::kDouble).reshape({n, cols}).contiguous();
        at::Tensor gx = need_x ? at::empty_like(x) : at::empty({0}, x.options());
        if (n == 0 || !need_x) return {gx, at::empty({0}, x.options())};
        hipStream_t stream = c10::hip::getCurrentHIPStream(x.get_device()).stream();
        std::lock_guard<std::mutex> lock(e->mu);
        sync_live(*e, x, ref_x, weights, biases, stream);
        check(molann_features_backward_f64(e->plan, x.data_ptr<double>(), g.data_ptr<double>(), n, gx.data_ptr<double>(), stream),
              "molann_features_backward_f64");
        return {gx, at::empty({0}, x.options())};
    }
    at::Tensor g = grad_out.to(at::kFloat).reshape({n, cols}).contiguous();
    at::Tensor gx = need_x ? at::empty_like(x) : at::empty({0}, x.options());
    at::Tensor gp = need_params ? at::zeros({molann_plan_grad_params_size(e->plan)}, x.options()) : at::empty({0}, x.options());
    if (n == 0) {
        if (need_x) gx.zero_();
        return {gx, gp};
    }
    hipStream_t stream = c10::hip::getCurrentHIPStream(x.get_device()).stream();
    std::lock_guard<std::mutex> lock(e->mu);
    sync_live(*e, x, ref_x, weights, biases, stream);
    check(molann_backward_f32(e->plan, x.data_ptr<float>(), g.data_ptr<float>(), n, need_x ? gx.data_ptr<float>() : nullptr,
                              need_params ? gp.data_ptr<float>() : nullptr, stream),
          "molann_backward_f32");
    return {gx, gp};
}

// {out, grad_x}: the forward's outputs and the vector-Jacobian product for grad_out in ONE launch (molann_value_and_vjp_f32: the
// one-pass backward that also stores the outputs).  Parameters are data.  `into` (optional: {out, grad_x} of the right shapes)
// is written instead of fresh tensors - a caller at every MD step keeps its two buffers.
std::vector<at::Tensor> value_and_vjp_impl(const at::Tensor& x_in, const std::vector<int64_t>& desc, const at::Tensor& ref_x, const std::vector<at::Tensor>& weights,
                                           const std::vector<at::Tensor>& biases, const at::Tensor& grad_out, const std::vector<at::Tensor>& into);
std::vector<at::Tensor> value_and_vjp_hip(const at::Tensor& x_in, std::vector<int64_t> desc, const at::Tensor& ref_x, std::vector<at::Tensor> weights,
                                          std::vector<at::Tensor> biases, const at::Tensor& grad_out, std::vector<at::Tensor> into) {
    return value_and_vjp_impl(x_in, desc, ref_x, weights, biases, grad_out, into);
}
std::vector<at::Tensor> value_and_vjp_h_hip(const at::Tensor& x_in, int64_t handle, const at::Tensor& ref_x, std::vector<at::Tensor> weights,
                                            std::vector<at::Tensor> biases, const at::Tensor& grad_out, std::vector<at::Tensor> into) {
    std::shared_ptr<const std::vector<int64_t>> d;
    {
        std::lock_guard<std::mutex> lock(g_handle_mu);
        TORCH_CHECK(handle >= 0 && (size_t)handle < g_handles.size(), "molann::value_and_vjp_h: unknown handle ", handle);
        d = g_handles[(size_t)handle];
    }
    return value_and_vjp_impl(x_in, *d, ref_x, weights, biases, grad_out, into);
}
std::vector<at::Tensor> value_and_vjp_impl(const at::Tensor& x_in, const std::vector<int64_t>& desc, const at::Tensor& ref_x, const std::vector<at::Tensor>& weights,
                                           const std::vector<at::Tensor>& biases, const at::Tensor& grad_out, const std::vector<at::Tensor>& into) {
    check_x(x_in, desc);
    TORCH_CHECK(x_in.scalar_type() == at::kFloat, "molann::value_and_vjp: float32 only");
    const at::Tensor x = x_in.contiguous();
    const c10::DeviceGuard guard(x.device());
    const bool align_only = desc[1] == KIND_ALIGN;
    auto e = entry_for(align_only ? align_as_features(desc) : desc, x, ref_x);
    const int64_t n = x.size(0);
    const int64_t cols = e->kind == KIND_FORWARD ? e->out_dim : e->feature_dim;
    at::Tensor g = grad_out.scalar_type() == at::kFloat && grad_out.is_contiguous() ? grad_out : grad_out.to(at::kFloat).contiguous();
    TORCH_CHECK(g.numel() == n * cols && g.device() == x.device(), "molann::value_and_vjp: grad_out must be [", n, ", ", cols, "] on ", x.device());
    at::Tensor out, gx;
    if (into.size() == 2) {
        out = into[0]; gx = into[1];
        TORCH_CHECK(out.is_contiguous() && gx.is_contiguous() && out.scalar_type() == at::kFloat && gx.scalar_type() == at::kFloat &&
                    out.numel() == n * cols && gx.numel() == x.numel() && out.device() == x.device() && gx.device() == x.device(),
                    "molann::value_and_vjp: `into` must be contiguous float32 {[N, out_dim], [N, n_inp, 3]} on x's device");
    } else {
        out = at::empty({n, cols}, x.options());
        gx = at::empty_like(x);
    }
    if (n == 0) return {out, gx};
    hipStream_t stream = c10::hip::getCurrentHIPStream(x.get_device()).stream();
    std::lock_guard<std::mutex> lock(e->mu);
    sync_live(*e, x, ref_x, weights, biases, stream);
    check(molann_value_and_vjp_f32(e->plan, x.data_ptr<float>(), g.data_ptr<float>(), n, out.data_ptr<float>(), gx.data_ptr<float>(), stream),
          "molann_value_and_vjp_f32");
    return {out, gx};
}

// The fused forward that also keeps the features: {out, features} - or {out, empty} where the plan has no such twin of its
// kernel (molann_plan_backward_kind != 1 ... != 2 plans recompute in molann_backward_f32).  float32 fused plans.
std::vector<at::Tensor> run_train_hip(const at::Tensor& x_in, std::vector<int64_t> desc, const at::Tensor& ref_x, std::vector<at::Tensor> weights,
                                      std::vector<at::Tensor> biases) {
    check_x(x_in, desc);
    TORCH_CHECK(desc[1] == KIND_FORWARD && x_in.scalar_type() == at::kFloat, "molann::run_train: float32 forward plans only");
    const at::Tensor x = x_in.contiguous();
    const c10::DeviceGuard guard(x.device());
    auto e = entry_for(desc, x, ref_x);
    const int64_t n = x.size(0);
    at::Tensor out = at::empty({n, e->out_dim}, x.options());
    at::Tensor feat = at::empty({n, e->feature_dim}, x.options());
    if (n == 0) return {out, feat};
    hipStream_t stream = c10::hip::getCurrentHIPStream(x.get_device()).stream();
    std::lock_guard<std::mutex> lock(e->mu);
    sync_live(*e, x, ref_x, weights, biases, stream);
    const int rc = molann_forward_train_f32(e->plan, x.data_ptr<float>(), n, out.data_ptr<float>(), feat.data_ptr<float>(), stream);
    if (rc == MOLANN_E_UNSUPPORTED) {
        check(molann_forward_packed_f32(e->plan, x.data_ptr<float>(), n, out.data_ptr<float>(), stream), "molann_forward_packed_f32");
        return {out, at::empty({0}, x.options())};
    }
    check(rc, "molann_forward_train_f32");
    return {out, feat};
}

// flat parameter gradients (dW_l, db_l per layer) of the MLP alone, from the features run_train kept
at::Tensor run_backward_mlp_hip(const at::Tensor& feat, std::vector<int64_t> desc, const at::Tensor& ref_x, std::vector<at::Tensor> weights,
                                std::vector<at::Tensor> biases, const at::Tensor& grad_out) {
    TORCH_CHECK(desc.size() >= DESC_HEAD && desc[1] == KIND_FORWARD && feat.dim() == 2 && feat.scalar_type() == at::kFloat, "molann::run_backward_mlp: bad arguments");
    const at::Tensor f = feat.contiguous();
    const c10::DeviceGuard guard(f.device());
    auto e = entry_for(desc, f, ref_x);
    const int64_t n = f.size(0);
    TORCH_CHECK(f.size(1) == e->feature_dim, "molann::run_backward_mlp: features are [*, ", e->feature_dim, "], got ", f.sizes());
    at::Tensor g = grad_out.to(at::kFloat).reshape({n, e->out_dim}).contiguous();
    at::Tensor gp = at::zeros({molann_plan_grad_params_size(e->plan)}, f.options());
    if (n == 0) return gp;
    hipStream_t stream = c10::hip::getCurrentHIPStream(f.get_device()).stream();
    std::lock_guard<std::mutex> lock(e->mu);
    // (the parameters were packed by the forward of this step; a step that changed them in between repacks here)
    sync_live(*e, f, ref_x, weights, biases, stream);
    check(molann_mlp_backward_f32(e->plan, f.data_ptr<float>(), g.data_ptr<float>(), n, nullptr, gp.data_ptr<float>(), stream),
          "molann_mlp_backward_f32");
    return gp;
}

// {grad_x, flat parameter gradients} from the features run_train kept, for plans whose backward is the launches on kept features
// (molann_plan_backward_kind == 1: wave-per-frame preprocessing with a small head): the MLP's backward, then the preprocessing's
std::vector<at::Tensor> run_backward_kept_hip(const at::Tensor& x_in, const at::Tensor& feat, std::vector<int64_t> desc, const at::Tensor& ref_x,
                                              std::vector<at::Tensor> weights, std::vector<at::Tensor> biases, const at::Tensor& grad_out,
                                              bool need_x, bool need_params) {
    check_x(x_in, desc);
    TORCH_CHECK(desc[1] == KIND_FORWARD && x_in.scalar_type() == at::kFloat && feat.dim() == 2 && feat.scalar_type() == at::kFloat,
                "molann::run_backward_kept: float32 forward plans only");
    const at::Tensor x = x_in.contiguous();
    const at::Tensor f = feat.contiguous();
    const c10::DeviceGuard guard(x.device());
    auto e = entry_for(desc, x, ref_x);
    const int64_t n = x.size(0);
    TORCH_CHECK(f.size(0) == n && f.size(1) == e->feature_dim, "molann::run_backward_kept: features are [", n, ", ", e->feature_dim, "], got ", f.sizes());
    at::Tensor g = grad_out.to(at::kFloat).reshape({n, e->out_dim}).contiguous();
    at::Tensor gx = need_x ? at::empty_like(x) : at::empty({0}, x.options());
    at::Tensor gp = need_params ? at::zeros({molann_plan_grad_params_size(e->plan)}, x.options()) : at::empty({0}, x.options());
    if (n == 0) {
        if (need_x) gx.zero_();
        return {gx, gp};
    }
    at::Tensor gf = need_x ? at::empty_like(f) : at::empty({0}, f.options());
    hipStream_t stream = c10::hip::getCurrentHIPStream(x.get_device()).stream();
    std::lock_guard<std::mutex> lock(e->mu);
    sync_live(*e, x, ref_x, weights, biases, stream);
    check(molann_mlp_backward_f32(e->plan, f.data_ptr<float>(), g.data_ptr<float>(), n, need_x ? gf.data_ptr<float>() : nullptr,
                                  need_params ? gp.data_ptr<float>() : nullptr, stream),
          "molann_mlp_backward_f32");
    if (need_x)
        check(molann_features_backward_f32(e->plan, x.data_ptr<float>(), gf.data_ptr<float>(), n, gx.data_ptr<float>(), stream),
              "molann_features_backward_f32");
    return {gx, gp};
}

// molann_plan_backward_kind of the plan of `desc` on x's device: 2 one pass over x, 1 launches on kept features, 0 none
int64_t backward_kind(const at::Tensor& x, std::vector<int64_t> desc, const at::Tensor& ref_x) {
    check_x(x, desc);
    TORCH_CHECK(x.is_cuda(), "molann::backward_kind: x must be a device tensor");
    const c10::DeviceGuard guard(x.device());
    auto e = entry_for(desc, x, ref_x);
    std::lock_guard<std::mutex> lock(e->mu);
    return molann_plan_backward_kind(e->plan);
}

// 1 if the plan of `desc` on x's device has backward kernels (molann_plan_supports_backward), else 0
int64_t supports_backward(const at::Tensor& x, std::vector<int64_t> desc, const at::Tensor& ref_x) {
    check_x(x, desc);
    TORCH_CHECK(x.is_cuda(), "molann::supports_backward: x must be a device tensor");
    const c10::DeviceGuard guard(x.device());
    return molann_plan_supports_backward(entry_for(desc, x, ref_x)->plan) == 1 ? 1 : 0;
}

// name + geometry of the kernels the plan of (desc, device) launched last ("" before its first launch)
std::string launch_info(std::vector<int64_t> desc, int64_t device) {
    std::shared_ptr<Entry> e;
    {
        std::lock_guard<std::mutex> lock(g_cache_mu);
        auto it = g_cache.find(std::make_pair(desc, cache_device_key((int)device)));
        if (it == g_cache.end()) return "";
        e = it->second;
    }
    char buf[256];
    buf[0] = 0;
    std::lock_guard<std::mutex> lock(e->mu);
    molann_plan_last_launch_info(e->plan, buf, (int)sizeof(buf));
    return buf;
}

// every cached plan derived from this description on this device (the plan itself, its features-only and
// alignment-as-features variants share the head up to the instance id)
template <typename F>
void for_plans_of(const std::vector<int64_t>& desc, int64_t device, F f) {
    std::lock_guard<std::mutex> lock(g_cache_mu);
    for (auto it = g_cache.begin(); it != g_cache.end();) {
        const std::vector<int64_t>& d = it->first.first;
        const bool same_dev = (it->first.second & 0xffff) == (int)device;
        const bool same_model = d.size() >= DESC_HEAD && desc.size() >= DESC_HEAD && d[9] == desc[9] && d[2] == desc[2] &&
                                (d == desc || desc[9] != 0);
        if (same_dev && same_model) it = f(it);
        else ++it;
    }
}

// the live tensors of this model were written in a way their version counters do not show (`.data`): repack
void invalidate(std::vector<int64_t> desc, int64_t device) {
    for_plans_of(desc, device, [](std::map<CacheKey, std::shared_ptr<Entry>>::iterator it) {
        std::lock_guard<std::mutex> lock(it->second->mu);
        it->second->dirty = true;
        return ++it;
    });
}

// the model is gone: give its plans (device blobs, workspaces, code objects) back - except those a captured graph still
// launches through (they go when the graph unpins them and the LRU gets to them)
void release(std::vector<int64_t> desc, int64_t device) {
    for_plans_of(desc, device, [](std::map<CacheKey, std::shared_ptr<Entry>>::iterator it) {
        return it->second->pins > 0 ? ++it : g_cache.erase(it);
    });
}

// A HIP graph captured through molann::run holds raw pointers to the plan's packed weights, reference and code objects:
// the plans of this model stay in the cache while pinned.  Returns how many plans were pinned / unpinned.
int64_t pin(std::vector<int64_t> desc, int64_t device) {
    int64_t n = 0;
    for_plans_of(desc, device, [&n](std::map<CacheKey, std::shared_ptr<Entry>>::iterator it) { ++it->second->pins; ++n; return ++it; });
    return n;
}
int64_t unpin(std::vector<int64_t> desc, int64_t device) {
    int64_t n = 0;
    for_plans_of(desc, device, [&n](std::map<CacheKey, std::shared_ptr<Entry>>::iterator it) {
        if (it->second->pins > 0) { --it->second->pins; ++n; }
        return ++it;
    });
    { std::lock_guard<std::mutex> lock(g_cache_mu); evict_lru_locked(); }
    return n;
}

int64_t drop_plans() {
    std::lock_guard<std::mutex> lock(g_cache_mu);
    const int64_t n = (int64_t)g_cache.size();
    g_cache.clear();
    return n;
}

int64_t cached_plans() {
    std::lock_guard<std::mutex> lock(g_cache_mu);
    return (int64_t)g_cache.size();
}

at::Tensor call_run(const at::Tensor& x, const std::vector<int64_t>& desc, const at::Tensor& ref_x,
                    const std::vector<at::Tensor>& weights, const std::vector<at::Tensor>& biases) {
    static auto op = c10::Dispatcher::singleton()
                         .findSchemaOrThrow("molann::run", "")
                         .typed<at::Tensor(const at::Tensor&, std::vector<int64_t>, const at::Tensor&, std::vector<at::Tensor>,
                                           std::vector<at::Tensor>)>();
    return op.call(x, desc, ref_x, weights, biases);
}

std::vector<at::Tensor> call_run_backward(const at::Tensor& x, const std::vector<int64_t>& desc, const at::Tensor& ref_x,
                                          const std::vector<at::Tensor>& weights, const std::vector<at::Tensor>& biases,
                                          const at::Tensor& grad_out, bool need_x, bool need_params) {
    static auto op = c10::Dispatcher::singleton()
                         .findSchemaOrThrow("molann::run_backward", "")
                         .typed<std::vector<at::Tensor>(const at::Tensor&, std::vector<int64_t>, const at::Tensor&,
                                                        std::vector<at::Tensor>, std::vector<at::Tensor>, const at::Tensor&, bool, bool)>();
    return op.call(x, desc, ref_x, weights, biases, grad_out, need_x, need_params);
}

std::vector<at::Tensor> call_run_train(const at::Tensor& x, const std::vector<int64_t>& desc, const at::Tensor& ref_x,
                                       const std::vector<at::Tensor>& weights, const std::vector<at::Tensor>& biases) {
    static auto op = c10::Dispatcher::singleton()
                         .findSchemaOrThrow("molann::run_train", "")
                         .typed<std::vector<at::Tensor>(const at::Tensor&, std::vector<int64_t>, const at::Tensor&, std::vector<at::Tensor>,
                                                        std::vector<at::Tensor>)>();
    return op.call(x, desc, ref_x, weights, biases);
}

at::Tensor call_run_backward_mlp(const at::Tensor& feat, const std::vector<int64_t>& desc, const at::Tensor& ref_x,
                                 const std::vector<at::Tensor>& weights, const std::vector<at::Tensor>& biases, const at::Tensor& grad_out) {
    static auto op = c10::Dispatcher::singleton()
                         .findSchemaOrThrow("molann::run_backward_mlp", "")
                         .typed<at::Tensor(const at::Tensor&, std::vector<int64_t>, const at::Tensor&, std::vector<at::Tensor>, std::vector<at::Tensor>,
                                           const at::Tensor&)>();
    return op.call(feat, desc, ref_x, weights, biases, grad_out);
}

std::vector<at::Tensor> call_run_backward_kept(const at::Tensor& x, const at::Tensor& feat, const std::vector<int64_t>& desc, const at::Tensor& ref_x,
                                               const std::vector<at::Tensor>& weights, const std::vector<at::Tensor>& biases, const at::Tensor& grad_out,
                                               bool need_x, bool need_params) {
    static auto op = c10::Dispatcher::singleton()
                         .findSchemaOrThrow("molann::run_backward_kept", "")
                         .typed<std::vector<at::Tensor>(const at::Tensor&, const at::Tensor&, std::vector<int64_t>, const at::Tensor&, std::vector<at::Tensor>,
                                                        std::vector<at::Tensor>, const at::Tensor&, bool, bool)>();
    return op.call(x, feat, desc, ref_x, weights, biases, grad_out, need_x, need_params);
}

std::vector<int64_t> features_only(const std::vector<int64_t>& desc);
at::Tensor activation(int64_t code, const at::Tensor& t);

// ---- create_graph=True (round 3) ---------------------------------------------------------------------------------------------
// The kernels' gradients carry no graph.  When a backward runs under an enabled grad mode (the caller wants to differentiate the
// gradients again: a loss on forces; the reference gets that from autograd through its SVD, ann.py:188-197) the gradient is
// rebuilt as a DIFFERENTIABLE composition: the float64 features as a node (Features64Fn) whose backward - J(x)^T g, the float64
// kernel - is itself a node (FeatBackward64Fn) with a backward of its own, and the MLP as ATen ops on the live parameters.
// FeatBackward64Fn's backward needs, for a cotangent v on J^T g, d/dx [v . J(x)^T g] and d/dg [..] = J(x) v: directional
// derivatives along v of the first-order kernel's output and of the features, taken as central differences of the float64
// kernels, per frame with h = 6e-6 max(1, |x|_max) / |v|_max (molann_amd/ann.py: _FeatBackward64 is the same in Python).
struct FeatBackward64Fn : public torch::autograd::Function<FeatBackward64Fn> {
    static at::Tensor forward(torch::autograd::AutogradContext* ctx, const at::Tensor& x, const at::Tensor& g, std::vector<int64_t> desc,
                              const at::Tensor& ref_x) {
        at::AutoDispatchBelowADInplaceOrView below;
        ctx->save_for_backward({x, g, ref_x});
        ctx->saved_data["desc"] = desc;
        return call_run_backward(x, desc, ref_x, {}, {}, g, true, false)[0];
    }
    static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx, torch::autograd::variable_list grad_outputs) {
        TORCH_CHECK(!at::GradMode::is_enabled(), "molann::run: gradients of order three are not available (the double backward is first-order itself)");
        const auto saved = ctx->get_saved_variables();
        const at::Tensor &x = saved[0], &g = saved[1], &ref_x = saved[2];
        const std::vector<int64_t> desc = ctx->saved_data["desc"].toIntVector();
        at::AutoDispatchBelowADInplaceOrView below;
        const at::Tensor v = grad_outputs[0].to(at::kDouble).contiguous();
        const at::Tensor vmax = v.abs().amax({1, 2}, true), xmax = x.abs().amax({1, 2}, true).clamp_min(1.0);
        const at::Tensor h = at::where(vmax > 0, 6e-6 * xmax / vmax.clamp_min(1e-300), at::zeros_like(vmax));
        const at::Tensor inv = at::where(h > 0, 0.5 / h.clamp_min(1e-300), at::zeros_like(h));
        const at::Tensor xp = (x + h * v).contiguous(), xm = (x - h * v).contiguous();
        at::Tensor gx, gg;
        if (ctx->needs_input_grad(0))
            gx = (call_run_backward(xp, desc, ref_x, {}, {}, g, true, false)[0] - call_run_backward(xm, desc, ref_x, {}, {}, g, true, false)[0]) * inv;
        if (ctx->needs_input_grad(1))
            gg = (call_run(xp, desc, ref_x, {}, {}) - call_run(xm, desc, ref_x, {}, {})) * inv.view({-1, 1});
        return {gx, gg, at::Tensor(), at::Tensor()};
    }
};

struct Features64Fn : public torch::autograd::Function<Features64Fn> {
    static at::Tensor forward(torch::autograd::AutogradContext* ctx, const at::Tensor& x, std::vector<int64_t> desc, const at::Tensor& ref_x) {
        at::AutoDispatchBelowADInplaceOrView below;
        ctx->save_for_backward({x, ref_x});
        ctx->saved_data["desc"] = desc;
        return call_run(x, desc, ref_x, {}, {});
    }
    static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx, torch::autograd::variable_list grad_outputs) {
        const auto saved = ctx->get_saved_variables();
        const std::vector<int64_t> desc = ctx->saved_data["desc"].toIntVector();
        const at::Tensor g = grad_outputs[0].to(at::kDouble).contiguous();
        if (at::GradMode::is_enabled()) return {FeatBackward64Fn::apply(saved[0], g, desc, saved[1]), at::Tensor(), at::Tensor()};
        at::AutoDispatchBelowADInplaceOrView below;
        return {call_run_backward(saved[0], desc, saved[1], {}, {}, g, true, false)[0], at::Tensor(), at::Tensor()};
    }
};

// the gradients of a RunFunction node as a differentiable composition: [x, ref_x, weights..., biases...]
torch::autograd::variable_list double_backward(const at::Tensor& x, const std::vector<int64_t>& desc, const at::Tensor& ref_x,
                                               const std::vector<at::Tensor>& weights, const std::vector<at::Tensor>& biases,
                                               const at::Tensor& grad_out, const std::vector<bool>& need) {
    const int64_t nl = (int64_t)weights.size();
    const std::vector<int64_t> fdesc = desc[1] == KIND_FORWARD ? features_only(desc) : (desc[1] == KIND_ALIGN ? align_as_features(desc) : desc);
    const at::Tensor ref64 = ref_x.numel() > 0 ? ref_x.detach().to(at::kDouble) : ref_x;
    at::Tensor h = Features64Fn::apply(x.to(at::kDouble), fdesc, ref64).to(x.scalar_type());
    if (desc[1] == KIND_ALIGN) h = h.view_as(x);
    if (desc[1] == KIND_FORWARD)
        for (int64_t l = 0; l < nl; ++l) {
            h = at::linear(h, weights[l], biases[l]);
            if (l + 1 < nl) h = activation(desc[7], h);
        }
    std::vector<at::Tensor> inputs;
    std::vector<int64_t> where;
    if (need[0]) { inputs.push_back(x); where.push_back(0); }
    for (int64_t l = 0; l < nl; ++l) if (need[2 + l]) { inputs.push_back(weights[l]); where.push_back(3 + l); }
    for (int64_t l = 0; l < nl; ++l) if (need[2 + nl + l]) { inputs.push_back(biases[l]); where.push_back(3 + nl + l); }
    torch::autograd::variable_list out(3 + 2 * nl);
    if (inputs.empty()) return out;
    const auto got = torch::autograd::grad({h}, inputs, {grad_out.to(h.scalar_type())}, /*retain_graph=*/true, /*create_graph=*/true, /*allow_unused=*/true);
    for (size_t i = 0; i < got.size(); ++i) out[where[i]] = got[i];
    return out;
}

// forward = one launch of the plan, nothing but the inputs saved; backward = molann_backward_f32, which
// recomputes the forward per frame (first-order only: the backward is not itself differentiable)
struct RunFunction : public torch::autograd::Function<RunFunction> {
    // apply() records one graph edge per at::Tensor / at::TensorList element and none for `desc`
    static at::Tensor forward(torch::autograd::AutogradContext* ctx, const at::Tensor& x, std::vector<int64_t> desc,
                              const at::Tensor& ref_x, at::TensorList weights, at::TensorList biases) {
        at::AutoDispatchBelowADInplaceOrView below;
        // What the backward will need is known here.  Gradients for x: molann_backward_f32 is one pass over x that
        // recomputes everything, nothing to keep.  Parameters only (x is data): the MLP's backward alone, on the features
        // this forward keeps - no second pass over x.
        at::Tensor out, feat;
        // (x wants a gradient too: kept features still pay where the backward is launches on them anyway - backward kind 1, a small
        // head behind the wave-per-frame kernels - instead of a recompute of the features inside molann_backward_f32)
        bool keep = desc.size() >= DESC_HEAD && desc[1] == KIND_FORWARD && x.scalar_type() == at::kFloat && x.is_cuda();
        if (keep && x.requires_grad()) keep = x.size(0) > 0 && backward_kind(x, desc, ref_x) == 1;
        if (keep) {
            std::vector<at::Tensor> r = call_run_train(x, desc, ref_x, weights.vec(), biases.vec());
            out = r[0];
            if (r[1].numel() > 0 || x.size(0) == 0) feat = r[1];
        } else {
            out = call_run(x, desc, ref_x, weights.vec(), biases.vec());
        }
        std::vector<at::Tensor> saved = {x, ref_x};
        for (auto& w : weights) saved.push_back(w);
        for (auto& b : biases) saved.push_back(b);
        if (feat.defined()) saved.push_back(feat);
        ctx->save_for_backward(saved);
        ctx->saved_data["desc"] = desc;
        ctx->saved_data["n_layers"] = (int64_t)weights.size();
        ctx->saved_data["kept_features"] = feat.defined();
        return out;
    }

    static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx, torch::autograd::variable_list grad_outputs) {
        // grad mode is on inside a backward only under create_graph=True: the caller wants to differentiate these gradients
        // again.  molann_backward_f32 is a kernel, its result has no graph: the gradient is then rebuilt as a differentiable
        // composition (double_backward above).
        if (at::GradMode::is_enabled()) {
            const auto sv = ctx->get_saved_variables();
            const int64_t n = ctx->saved_data["n_layers"].toInt();
            std::vector<at::Tensor> w(sv.begin() + 2, sv.begin() + 2 + n), b(sv.begin() + 2 + n, sv.begin() + 2 + 2 * n);
            std::vector<bool> need(2 + 2 * n);
            for (int64_t i = 0; i < 2 + 2 * n; ++i) need[(size_t)i] = ctx->needs_input_grad((size_t)i);
            return double_backward(sv[0], ctx->saved_data["desc"].toIntVector(), sv[1], w, b, grad_outputs[0], need);
        }
        const auto saved = ctx->get_saved_variables();
        const std::vector<int64_t> desc = ctx->saved_data["desc"].toIntVector();
        const int64_t nl = ctx->saved_data["n_layers"].toInt();
        const at::Tensor& x = saved[0];
        const at::Tensor& ref_x = saved[1];
        std::vector<at::Tensor> weights(saved.begin() + 2, saved.begin() + 2 + nl), biases(saved.begin() + 2 + nl, saved.begin() + 2 + 2 * nl);
        // graph edges (tensors only): x, ref_x, weights..., biases...; returned list (all inputs): x, desc, ref_x, ...
        const bool need_x = ctx->needs_input_grad(0);
        bool need_p = false;
        for (int64_t i = 0; i < 2 * nl; ++i) need_p = need_p || ctx->needs_input_grad(2 + i);
        std::vector<at::Tensor> g;
        {
            at::AutoDispatchBelowADInplaceOrView below;
            if (ctx->saved_data["kept_features"].toBool() && !need_x) {
                g = {at::Tensor(), need_p ? call_run_backward_mlp(saved[2 + 2 * nl], desc, ref_x, weights, biases, grad_outputs[0]) : at::Tensor()};
            } else if (ctx->saved_data["kept_features"].toBool() && backward_kind(x, desc, ref_x) == 1) {
                g = call_run_backward_kept(x, saved[2 + 2 * nl], desc, ref_x, weights, biases, grad_outputs[0], need_x, need_p);
            } else {
                g = call_run_backward(x, desc, ref_x, weights, biases, grad_outputs[0], need_x, need_p);
            }
        }
        torch::autograd::variable_list out(3 + 2 * nl);
        if (need_x) out[0] = g[0];
        if (need_p) {
            int64_t off = 0;
            for (int64_t l = 0; l < nl; ++l) { // flat layout: dW_l[J][K] then db_l[J], layer after layer
                const int64_t nw = weights[l].numel(), nb = biases[l].numel();
                if (ctx->needs_input_grad(2 + l)) out[3 + l] = g[1].narrow(0, off, nw).view(weights[l].sizes());
                if (ctx->needs_input_grad(2 + nl + l)) out[3 + nl + l] = g[1].narrow(0, off + nw, nb).view(biases[l].sizes());
                off += nw + nb;
            }
        }
        return out;
    }
};

// the plan of `desc` without its MLP (what PreprocessingANN.forward launches)
std::vector<int64_t> features_only(const std::vector<int64_t>& desc) {
    const int64_t n_layers = desc[6];
    std::vector<int64_t> v(desc.begin(), desc.end() - (n_layers > 0 ? n_layers + 1 : 0));
    v[1] = KIND_FEATURES;
    v[6] = 0;
    return v;
}

at::Tensor activation(int64_t code, const at::Tensor& t) {
    switch (code) {
    case MOLANN_ACT_TANH: return at::tanh(t);
    case MOLANN_ACT_RELU: return at::relu(t);
    case MOLANN_ACT_SIGMOID: return at::sigmoid(t);
    case MOLANN_ACT_IDENTITY: return t;
    case MOLANN_ACT_ELU: return at::elu(t);
    case MOLANN_ACT_SILU: return at::silu(t);
    case MOLANN_ACT_SOFTPLUS: return at::softplus(t);
    case MOLANN_ACT_LEAKY_RELU: return at::leaky_relu(t, 0.01);
    case MOLANN_ACT_GELU: return at::gelu(t);
    default: TORCH_CHECK(false, "molann::run: unknown activation code ", code);
    }
}

at::Tensor run_autograd(const at::Tensor& x, std::vector<int64_t> desc, const at::Tensor& ref_x, std::vector<at::Tensor> weights,
                        std::vector<at::Tensor> biases) {
    if (x.scalar_type() == at::kDouble) {
        // float64 (`model.double()`): gradients for the preprocessing come from molann_features_backward_f64; the MLP of a
        // float64 model under grad mode is ATen's (its parameters are read as they are anyway)
        bool needs = at::GradMode::is_enabled() && x.requires_grad();
        for (const auto& w : weights) needs = needs || (at::GradMode::is_enabled() && w.requires_grad());
        for (const auto& b : biases) needs = needs || (at::GradMode::is_enabled() && b.requires_grad());
        if (!needs) {
            at::AutoDispatchBelowADInplaceOrView below;
            return call_run(x, desc, ref_x, weights, biases);
        }
        if (desc.size() >= DESC_HEAD && desc[1] == KIND_FORWARD) {
            at::Tensor h = call_run(x, features_only(desc), ref_x, {}, {});
            for (size_t l = 0; l < weights.size(); ++l) {
                h = at::linear(h, weights[l], biases[l]);
                if (l + 1 < weights.size()) h = activation(desc[7], h);
            }
            return h;
        }
        return RunFunction::apply(x, desc, ref_x, at::TensorList(weights), at::TensorList(biases));
    }
    // A fused plan without a backward kernel (MLP wider than 32, large frames, ELU / GELU / Softplus) that has to
    // record gradients: features and their gradient from the kernels (the plan without the MLP), the MLP as ATen
    // ops - the composition molann_amd/ann.py: MolANN.forward uses in the same situation.
    if (desc.size() >= DESC_HEAD && desc[1] == KIND_FORWARD && at::GradMode::is_enabled() && x.is_cuda()) {
        bool needs = x.requires_grad();
        for (const auto& w : weights) needs = needs || w.requires_grad();
        for (const auto& b : biases) needs = needs || b.requires_grad();
        if (needs) {
            check_x(x, desc);
            bool fused_backward;
            {
                const c10::DeviceGuard guard(x.device());
                fused_backward = molann_plan_supports_backward(entry_for(desc, x, ref_x)->plan) == 1;
            }
            if (!fused_backward) {
                at::Tensor h = call_run(x, features_only(desc), ref_x, {}, {});
                for (size_t l = 0; l < weights.size(); ++l) {
                    h = at::linear(h, weights[l], biases[l]);
                    if (l + 1 < weights.size()) h = activation(desc[7], h);
                }
                return h;
            }
        }
    }
    return RunFunction::apply(x, desc, ref_x, at::TensorList(weights), at::TensorList(biases));
}

} // namespace

TORCH_LIBRARY(molann, m) {
    m.def("run(Tensor x, int[] desc, Tensor ref_x, Tensor[] weights, Tensor[] biases) -> Tensor");
    m.def("run_backward(Tensor x, int[] desc, Tensor ref_x, Tensor[] weights, Tensor[] biases, Tensor grad_out, bool need_x, "
          "bool need_params) -> Tensor[]");
    m.def("run_train(Tensor x, int[] desc, Tensor ref_x, Tensor[] weights, Tensor[] biases) -> Tensor[]");
    m.def("run_backward_mlp(Tensor feat, int[] desc, Tensor ref_x, Tensor[] weights, Tensor[] biases, Tensor grad_out) -> Tensor");
    m.def("run_backward_kept(Tensor x, Tensor feat, int[] desc, Tensor ref_x, Tensor[] weights, Tensor[] biases, Tensor grad_out, bool need_x, "
          "bool need_params) -> Tensor[]");
    m.def("backward_kind(Tensor x, int[] desc, Tensor ref_x) -> int", backward_kind);
    m.def("register_desc(int[] desc) -> int", register_desc);
    m.def("run_h(Tensor x, int handle, Tensor ref_x, Tensor[] weights, Tensor[] biases) -> Tensor");
    m.def("value_and_vjp_h(Tensor x, int handle, Tensor ref_x, Tensor[] weights, Tensor[] biases, Tensor grad_out, Tensor[] into) -> Tensor[]");
    m.def("value_and_vjp(Tensor x, int[] desc, Tensor ref_x, Tensor[] weights, Tensor[] biases, Tensor grad_out, Tensor[] into) -> Tensor[]");
    m.def("supports_backward(Tensor x, int[] desc, Tensor ref_x) -> int", supports_backward);
    m.def("launch_info(int[] desc, int device) -> str", launch_info);
    m.def("invalidate(int[] desc, int device) -> ()", invalidate);
    m.def("release(int[] desc, int device) -> ()", release);
    m.def("pin(int[] desc, int device) -> int", pin);
    m.def("unpin(int[] desc, int device) -> int", unpin);
    m.def("drop_plans() -> int", drop_plans);
    m.def("cached_plans() -> int", cached_plans);
}

TORCH_LIBRARY_IMPL(molann, CUDA, m) { // ROCm builds of torch name the HIP device "cuda"
    m.impl("run", run_hip);
    m.impl("run_backward", run_backward_hip);
    m.impl("run_train", run_train_hip);
    m.impl("run_backward_mlp", run_backward_mlp_hip);
    m.impl("run_backward_kept", run_backward_kept_hip);
    m.impl("value_and_vjp", value_and_vjp_hip);
    m.impl("run_h", run_h_hip);
    m.impl("value_and_vjp_h", value_and_vjp_h_hip);
}

TORCH_LIBRARY_IMPL(molann, Autograd, m) { m.impl("run", run_autograd); }
