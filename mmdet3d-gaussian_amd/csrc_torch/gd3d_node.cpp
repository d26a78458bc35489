// gd3d_node.cpp — the autograd node of a reduced GDLoss call as a C++ torch::autograd::Function.
//
// Host layer ABOVE the C ABI (include/gd3d.h): it owns no arithmetic.  It exists because a Python autograd.Function
// costs ~24 us of interpreter and engine time per forward+backward on the GPU boxes' hosts before the first launch is made
// (profiles/r02_small_p_latency.jsonl, FLOOR row), which is most of a training-size GDLoss call (P <= 1e4 positives, ~10 us
// of GPU work).  The node does what mmdet3d-gaussian_amd/gd_loss.py::_GDReduced does — allocate outputs, one
// gd3d_loss_fused_* call in forward, one gd3d_grad_finish call in backward — with the same semantics
// (/root/reference/mmdet3d_gaussian/models/losses/gaussian_distance_loss.py:280-310 through the kernels).
//
// No HIP or CUDA header is included: tensors are allocated through ATen's dispatcher, the raw stream handle is passed in
// from Python (torch._C._cuda_getCurrentRawStream) and reused in backward (the autograd engine runs a node's backward on
// the stream its forward ran on), and libgd3d.so is reached through dlopen (the process has it loaded via ctypes already).
#include <torch/extension.h>

#include <dlfcn.h>

#include "../../include/gd3d.h"

namespace {

struct Api {
  decltype(&gd3d_loss_workspace_bytes) ws_bytes = nullptr;
  decltype(&gd3d_loss_fused_timed) fused = nullptr;
  decltype(&gd3d_loss_fused_select) select = nullptr;
  decltype(&gd3d_grad_finish) finish = nullptr;
  decltype(&gd3d_abi_version) abi = nullptr;
} api;

void bind_library(const std::string& path) {
  void* h = dlopen(path.c_str(), RTLD_NOW | RTLD_GLOBAL);
  TORCH_CHECK(h != nullptr, "gd3d node: cannot load ", path, ": ", dlerror());
  auto sym = [&](const char* name) {
    void* p = dlsym(h, name);
    TORCH_CHECK(p != nullptr, "gd3d node: ", path, " does not export ", name);
    return p;
  };
  api.ws_bytes = reinterpret_cast<decltype(api.ws_bytes)>(sym("gd3d_loss_workspace_bytes"));
  api.fused = reinterpret_cast<decltype(api.fused)>(sym("gd3d_loss_fused_timed"));
  api.select = reinterpret_cast<decltype(api.select)>(sym("gd3d_loss_fused_select"));
  api.finish = reinterpret_cast<decltype(api.finish)>(sym("gd3d_grad_finish"));
  api.abi = reinterpret_cast<decltype(api.abi)>(sym("gd3d_abi_version"));
  TORCH_CHECK(api.abi(nullptr) == GD3D_ABI_VERSION, "gd3d node: ABI version mismatch");
}

// GDLoss hyper-parameters (+ optional fused bbox-coder prologue) of one module configuration, built once in Python
struct NodeParams {
  gd3d_params p;
  bool has_pro = false;
  gd3d_prologue pro;
  at::Tensor aux;  // keeps the prologue's device array alive
  NodeParams(int64_t loss_type, int64_t fun, double tau, double alpha, double c0, double c1, double c2, int64_t flag) {
    p.loss_type = (int32_t)loss_type;
    p.fun = (int32_t)fun;
    p.tau = (float)tau;
    p.alpha = (float)alpha;
    p.center_offset[0] = (float)c0;
    p.center_offset[1] = (float)c1;
    p.center_offset[2] = (float)c2;
    p.flag = (int32_t)flag;
  }
  void set_prologue(int64_t kind, bool norm_bbox, const at::Tensor& aux_, double osf, double vs0, double vs1, double pc0,
                    double pc1) {
    has_pro = true;
    aux = aux_;
    pro.kind = (int32_t)kind;
    pro.norm_bbox = norm_bbox ? 1 : 0;
    pro.aux = aux.data_ptr<float>();
    pro.out_size_factor = (float)osf;
    pro.voxel_size[0] = (float)vs0;
    pro.voxel_size[1] = (float)vs1;
    pro.pc_range[0] = (float)pc0;
    pro.pc_range[1] = (float)pc1;
    pro.reserved = 0.0f;
  }
};

void check_rc(int rc, const char* what) { TORCH_CHECK(rc == 0, what, " failed with code ", rc); }

struct Launch {
  at::Tensor total, any_pos, gp, gt;
};

Launch run_forward(const at::Tensor& pred, const at::Tensor& target, const at::Tensor& weight, const NodeParams& np,
                   double scale, bool select, bool want_sum, bool need_gp, bool need_gt, int64_t stream, int64_t ev0,
                   int64_t ev1) {
  const int64_t n = pred.size(0);
  const auto opt = pred.options();
  Launch L;
  at::Tensor ws;
  if (want_sum) {
    L.total = at::empty({}, opt);
    ws = at::empty({(int64_t)(api.ws_bytes(n) / 4)}, opt);
  }
  if (need_gp) L.gp = at::empty_like(pred);
  if (need_gt) L.gt = at::empty_like(target);
  float* gp = need_gp ? L.gp.data_ptr<float>() : nullptr;
  float* gt = need_gt ? L.gt.data_ptr<float>() : nullptr;
  const gd3d_prologue* pro = np.has_pro ? &np.pro : nullptr;
  if (select) {
    L.any_pos = at::empty({1}, opt.dtype(at::kInt));
    check_rc(api.select(&np.p, pro, pred.data_ptr<float>(), target.data_ptr<float>(), weight.data_ptr<float>(), n,
                        (float)scale, L.total.data_ptr<float>(), L.any_pos.data_ptr<int32_t>(), gp, gt, ws.data_ptr(),
                        (void*)stream, (void*)ev0, (void*)ev1),
             "gd3d_loss_fused_select");
  } else {
    const float* w1 = nullptr;
    const float* w7 = nullptr;
    if (weight.defined()) (weight.dim() == 2 ? w7 : w1) = weight.data_ptr<float>();
    check_rc(api.fused(&np.p, pro, pred.data_ptr<float>(), target.data_ptr<float>(), w1, w7, n, (float)scale, nullptr,
                       want_sum ? L.total.data_ptr<float>() : nullptr, gp, gt, want_sum ? ws.data_ptr() : nullptr,
                       (void*)stream, (void*)ev0, (void*)ev1),
             "gd3d_loss_fused");
  }
  return L;
}

// the node keeps its own copy of the parameters (a handful of scalars + the prologue's aux tensor) in saved_data: the
// Python-side NodeParams object may die before backward runs
c10::IValue pack(const NodeParams& np) {
  return c10::IValue(std::vector<double>{(double)np.p.loss_type, (double)np.p.fun, np.p.tau, np.p.alpha, np.p.center_offset[0],
                                         np.p.center_offset[1], np.p.center_offset[2], (double)np.p.flag,
                                         np.has_pro ? 1.0 : 0.0, (double)np.pro.kind, (double)np.pro.norm_bbox,
                                         np.pro.out_size_factor, np.pro.voxel_size[0], np.pro.voxel_size[1],
                                         np.pro.pc_range[0], np.pro.pc_range[1]});
}
NodeParams unpack(const c10::IValue& v, const at::Tensor& aux) {
  const auto d = v.toDoubleVector();
  NodeParams np((int64_t)d[0], (int64_t)d[1], d[2], d[3], d[4], d[5], d[6], (int64_t)d[7]);
  if (d[8] != 0.0) np.set_prologue((int64_t)d[9], d[10] != 0.0, aux, d[11], d[12], d[13], d[14], d[15]);
  return np;
}

class GDReducedNode : public torch::autograd::Function<GDReducedNode> {
 public:
  static torch::autograd::variable_list forward(torch::autograd::AutogradContext* ctx, const at::Tensor& pred,
                                                const at::Tensor& target, const at::Tensor& weight, const NodeParams* np,
                                                double scale, bool select, bool need_gp, bool need_gt, int64_t stream,
                                                int64_t ev0, int64_t ev1) {
    c10::DeviceGuard guard(pred.device());
    Launch L = run_forward(pred, target, weight, *np, scale, select, true, need_gp, need_gt, stream, ev0, ev1);
    auto& sd = ctx->saved_data;
    sd["pred"] = pred;
    sd["target"] = target;
    if (weight.defined()) sd["weight"] = weight;
    if (need_gp) sd["gp"] = L.gp;
    if (need_gt) sd["gt"] = L.gt;
    if (select) sd["any_pos"] = L.any_pos;
    sd["params"] = pack(*np);
    if (np->has_pro) sd["aux"] = np->aux;
    sd["scale"] = scale;
    sd["select"] = select;
    sd["stream"] = stream;
    sd["used"] = false;
    if (select) {
      ctx->mark_non_differentiable({L.any_pos});
      return {L.total, L.any_pos};
    }
    return {L.total};
  }

  static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx,
                                                 torch::autograd::variable_list grad_outputs) {
    auto& sd = ctx->saved_data;
    const at::Tensor pred = sd["pred"].toTensor(), target = sd["target"].toTensor();
    const at::Tensor weight = sd.count("weight") ? sd["weight"].toTensor() : at::Tensor();
    const NodeParams np = unpack(sd["params"], sd.count("aux") ? sd["aux"].toTensor() : at::Tensor());
    const bool select = sd["select"].toBool();
    const int64_t stream = sd["stream"].toInt();
    at::Tensor gp = sd.count("gp") ? sd["gp"].toTensor() : at::Tensor();
    at::Tensor gt = sd.count("gt") ? sd["gt"].toTensor() : at::Tensor();
    torch::autograd::variable_list out(11);
    if (!gp.defined() && !gt.defined()) return out;
    c10::DeviceGuard guard(pred.device());
    if (sd["used"].toBool()) {  // retain_graph replay: the saved buffers were scaled in place; recompute them
      Launch L = run_forward(pred, target, weight, np, sd["scale"].toDouble(), false, false, gp.defined(), gt.defined(),
                             stream, 0, 0);
      gp = L.gp;
      gt = L.gt;
    } else {
      sd["used"] = true;
    }
    at::Tensor g = grad_outputs[0];
    if (g.scalar_type() != at::kFloat) g = g.to(at::kFloat);
    const at::Tensor any_pos = select ? sd["any_pos"].toTensor() : at::Tensor();
    check_rc(api.finish(gp.defined() ? gp.data_ptr<float>() : nullptr, gt.defined() ? gt.data_ptr<float>() : nullptr,
                        g.data_ptr<float>(), pred.size(0), select ? any_pos.data_ptr<int32_t>() : nullptr,
                        select ? weight.data_ptr<float>() : nullptr,
                        (select && np.has_pro) ? pred.data_ptr<float>() : nullptr, (select && np.has_pro) ? &np.pro : nullptr,
                        (void*)stream),
             "gd3d_grad_finish");
    out[0] = gp;
    out[1] = gt;
    return out;
  }
};

// pred / target: contiguous fp32 (N,7) on the GPU; weight: None, (N,) or (N,7) fp32 (select requires (N,7)).
// Returns [loss_sum] or, with select, [loss_sum, any_positive].
std::vector<at::Tensor> gd_reduced(const at::Tensor& pred, const at::Tensor& target, const c10::optional<at::Tensor>& weight,
                                   const NodeParams& np, double scale, bool select, int64_t stream, int64_t ev0, int64_t ev1) {
  TORCH_CHECK(api.fused != nullptr, "gd3d node: bind_library() has not been called");
  TORCH_CHECK(pred.dim() == 2 && pred.size(1) == 7 && pred.scalar_type() == at::kFloat && pred.is_contiguous() &&
                  target.sizes() == pred.sizes() && target.scalar_type() == at::kFloat && target.is_contiguous(),
              "gd3d node: pred / target must be contiguous fp32 (N,7)");
  const at::Tensor w = weight.has_value() ? *weight : at::Tensor();
  const bool grad_on = at::GradMode::is_enabled();   // (inside forward() it is always off)
  return GDReducedNode::apply(pred, target, w, &np, scale, select, grad_on && pred.requires_grad(),
                              grad_on && target.requires_grad(), stream, ev0, ev1);
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.def("bind_library", &bind_library, "dlopen libgd3d.so and resolve the entry points the node calls");
  m.def("gd_reduced", &gd_reduced);
  py::class_<NodeParams>(m, "NodeParams")
      .def(py::init<int64_t, int64_t, double, double, double, double, double, int64_t>())
      .def("set_prologue", &NodeParams::set_prologue);
}
