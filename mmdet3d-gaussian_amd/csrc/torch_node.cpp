// torch_node.cpp — OPTIONAL host glue above the C ABI, in C++: the autograd node of GDLoss's reduced forms (reduction 'mean' / 'sum'),
// of the anchor-head slice and of scatter_reduce, and the allocate-launch-read-back sequence of nms_gpu's scored path (nms_scored, at
// the end).  The package's first-class host layer is mmdet3d-gaussian_amd/_pynode.py (torch.autograd.Function + ctypes, same function
// surface, bit-identical results); this module is an accelerator `_lib.load_node()` prefers when it builds (GD3D_HOST=python|cpp).
//
// Why this exists.  A training-size GDLoss call (64-4096 positives, gd_anchor3d_head.py:137-141) is one 6 us launch; as a
// Python torch.autograd.Function it cost 15 us in forward and 35-60 us per call under backward(), of which 3.8 / 23-41 us
// are what an EMPTY Python Function costs (profiles/r04_small_p_latency.jsonl): the engine thread has to take the GIL
// and walk the Python ctx for every node.  A C++ node is called by the engine directly.  This file is host plumbing above
// the C ABI of include/gd3d.h and nothing else: every launch goes through the same extern "C" entry points the ctypes
// layer binds, with raw pointers taken from the tensors; it contains no kernel and no arithmetic on box data.
//
// What it mirrors: the reduce / weight / early-out semantics of gaussian_distance_loss.py:280-310 + mmdet's
// weight_reduce_loss are decided in gd_loss.py (GDLoss.forward) and arrive here as `scale`, `row_weight` and `select`.
//   forward : ONE fused launch writes the loss sum AND the final gradients (csrc/gd3d_loss.hip); the gradients wait in the
//             node.
//   backward: hands them over — scaled by the upstream gradient on the device (gd3d_grad_finish; no host sync), or
//             untouched when the upstream gradient is the library's own constant 1.0 (gd_loss.unit_grad, known by address).
//             A second backward under retain_graph recomputes them.  Differentiating the result again raises (the
//             gradients were written by a kernel: there is no graph behind them), as torch's once_differentiable does.
#include <torch/extension.h>
#include <torch/csrc/autograd/functions/basic_ops.h>
#include <torch/csrc/autograd/functions/utils.h>
#include <torch/csrc/autograd/saved_variable.h>
#include <c10/hip/HIPStream.h>
#include <ATen/Parallel.h>

#include <dlfcn.h>

#include <atomic>
#include <chrono>
#include <cstring>
#include <mutex>

#include "../../include/gd3d.h"

namespace {

using at::Tensor;
using torch::autograd::variable_list;

// The entry points of include/gd3d.h this node calls, resolved from the library the ctypes layer loaded (`bind(path)`:
// dlopen of a loaded library returns the same image, so an experimental build selected with GD3D_LIB is the one used here too).
struct Abi {
  decltype(&::gd3d_loss_fused_decoded) loss_fused_decoded = nullptr;
  decltype(&::gd3d_loss_fused_timed) loss_fused_timed = nullptr;
  decltype(&::gd3d_loss_fused_select) loss_fused_select = nullptr;
  decltype(&::gd3d_loss_fused_one_launch) loss_fused_one_launch = nullptr;
  decltype(&::gd3d_grad_finish) grad_finish = nullptr;
  decltype(&::gd3d_loss_fused_cpu) loss_fused_cpu = nullptr;
  decltype(&::gd3d_scale_rows_cpu) scale_rows_cpu = nullptr;
  decltype(&::gd3d_abi_version) abi_version = nullptr;
  decltype(&::rnms_scored) nms_scored = nullptr;
  decltype(&::gd3d_anchor_head_bbox_loss) anchor_head = nullptr;
  decltype(&::gd3d_anchor_head_bbox_loss_dyn) anchor_head_dyn = nullptr;
  decltype(&::gd3d_scale_rows) scale_rows = nullptr;
  decltype(&::gd3d_loss_workspace_bytes) loss_workspace_bytes = nullptr;
  decltype(&::vox_scatter_reduce) scatter_reduce = nullptr;
  decltype(&::vox_scatter_backward) scatter_backward = nullptr;
  decltype(&::vox_scatter_backward_grouped) scatter_backward_grouped = nullptr;
  decltype(&::rnms_scored_workspace_bytes) nms_scored_workspace_bytes = nullptr;
  bool bound = false;
} abi;

template <typename F>
void resolve(void* image, const char* name, F& slot) {
  slot = reinterpret_cast<F>(dlsym(image, name));
  TORCH_CHECK(slot != nullptr, "gd3d node: libgd3d.so does not export ", name);
}

int bind(const std::string& path) {
  void* image = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);   // the image ctypes already mapped: same handle, nothing promoted
  TORCH_CHECK(image != nullptr, "gd3d node: cannot open ", path, ": ", dlerror());
  resolve(image, "gd3d_loss_fused_decoded", abi.loss_fused_decoded);
  resolve(image, "gd3d_loss_fused_timed", abi.loss_fused_timed);
  resolve(image, "gd3d_loss_fused_select", abi.loss_fused_select);
  resolve(image, "gd3d_loss_fused_one_launch", abi.loss_fused_one_launch);
  resolve(image, "gd3d_grad_finish", abi.grad_finish);
  resolve(image, "gd3d_loss_fused_cpu", abi.loss_fused_cpu);
  resolve(image, "gd3d_scale_rows_cpu", abi.scale_rows_cpu);
  resolve(image, "gd3d_abi_version", abi.abi_version);
  resolve(image, "rnms_scored", abi.nms_scored);
  resolve(image, "gd3d_anchor_head_bbox_loss", abi.anchor_head);
  resolve(image, "gd3d_anchor_head_bbox_loss_dyn", abi.anchor_head_dyn);
  resolve(image, "gd3d_scale_rows", abi.scale_rows);
  resolve(image, "gd3d_loss_workspace_bytes", abi.loss_workspace_bytes);
  resolve(image, "vox_scatter_reduce", abi.scatter_reduce);
  resolve(image, "vox_scatter_backward", abi.scatter_backward);
  resolve(image, "vox_scatter_backward_grouped", abi.scatter_backward_grouped);
  resolve(image, "rnms_scored_workspace_bytes", abi.nms_scored_workspace_bytes);
  const int version = abi.abi_version(nullptr);
  TORCH_CHECK(version == GD3D_ABI_VERSION, "gd3d node: built against ABI ", GD3D_ABI_VERSION, ", the library reports ", version);
  abi.bound = true;
  return version;
}

std::atomic<int64_t> g_finish_calls{0};   // gd3d_grad_finish launches made so far (tests: unit_grad must make none)

constexpr int MAX_DEVICES = 64;
const void* g_unit_grad[MAX_DEVICES + 1] = {};   // [MAX_DEVICES] = the CPU's

// the `_cpu` twins are built for x86-64-v3 (build.py): a clear error instead of SIGILL on a host without AVX2 / FMA
void require_cpu_twins() {
  static const bool ok = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma");
  TORCH_CHECK(ok, "GDLoss on CPU tensors: the CPU twins of libgd3d.so are built for x86-64-v3 (AVX2 + FMA) and this host CPU lacks them");
}

void fail(int rc, const char* what) {
  TORCH_CHECK(rc == 0, what, " failed with code ", rc,
              rc == GD3D_E_BADARG ? " (bad argument)" : rc == GD3D_E_HOST ? " (a host worker thread failed: out of memory?)" : "");
}

inline float* fp(const Tensor& t) { return t.defined() ? t.data_ptr<float>() : nullptr; }

void* current_stream(const Tensor& t) { return (void*)c10::hip::getCurrentHIPStream(t.device().index()).stream(); }

bool is_unit_grad(const Tensor& g) {
  const int slot = g.is_cuda() ? (int)g.device().index() : MAX_DEVICES;
  return slot >= 0 && slot <= MAX_DEVICES && g_unit_grad[slot] != nullptr && g.dim() == 0 &&
         g.scalar_type() == at::kFloat && g.data_ptr() == g_unit_grad[slot];
}

struct Weights {
  const float* w1 = nullptr;
  const float* w7 = nullptr;
  explicit Weights(const Tensor& w) {
    if (w.defined()) (w.dim() == 2 ? w7 : w1) = w.data_ptr<float>();
  }
};

struct ReducedBackward : public torch::autograd::Node {
  torch::autograd::SavedVariable pred_, target_;
  Tensor weight_, aux_, gp_, gt_, buf_;   // aux_: the tensor behind prologue_.aux; buf_: owns the any-positive flag
  gd3d_params params_;
  gd3d_prologue prologue_;
  bool has_prologue_ = false, select_ = false, used_ = false, want_gp_ = false, want_gt_ = false;
  float scale_ = 1.0f;
  int64_t n_ = 0;

  std::string name() const override { return "GDLossReducedBackward"; }

  void release_variables() override {
    std::lock_guard<std::mutex> lock(mutex_);
    pred_.reset_data();
    target_.reset_data();
    weight_.reset();
    aux_.reset();
    gp_.reset();
    gt_.reset();
    buf_.reset();
  }

  variable_list apply(variable_list&& grads) override {
    std::lock_guard<std::mutex> lock(mutex_);
    const bool twice = at::GradMode::is_enabled();   // create_graph=True: the only case with grad mode ON in here
    at::AutoGradMode no_grad(false);
    variable_list out(2);
    const Tensor pred = pred_.unpack(), target = target_.unpack();   // raises after a released graph
    const Tensor& g_in = grads[0];
    if (!g_in.defined() || (!want_gp_ && !want_gt_)) return out;
    Tensor gp, gt;
    const gd3d_prologue* pro = has_prologue_ ? &prologue_ : nullptr;
    const bool cuda = pred.is_cuda();
    c10::OptionalDeviceGuard device_guard;
    if (cuda) device_guard.reset_device(pred.device());
    if (used_) {   // retain_graph replay: the buffers of the first backward were handed over (and scaled in place)
      if (want_gp_) gp = at::empty_like(pred);
      if (want_gt_) gt = at::empty_like(target);
      const Weights w(weight_);
      if (cuda)
        fail(abi.loss_fused_decoded(&params_, pro, fp(pred), fp(target), w.w1, w.w7, n_, scale_, nullptr, nullptr, fp(gp),
                                     fp(gt), nullptr, current_stream(pred)), "gd3d_loss_fused_decoded");
      else
        fail(abi.loss_fused_cpu(&params_, fp(pred), fp(target), w.w1, w.w7, n_, scale_, nullptr, nullptr, fp(gp), fp(gt),
                                 nullptr, at::get_num_threads()), "gd3d_loss_fused_cpu");
    } else {   // hand the buffers over: with no reference left here a leaf's AccumulateGrad keeps them instead of cloning
      gp = std::move(gp_);
      gt = std::move(gt_);
      gp_ = Tensor();
      gt_ = Tensor();
      used_ = true;
    }
    if (select_ || !is_unit_grad(g_in)) {
      Tensor g = g_in.scalar_type() == at::kFloat ? g_in : g_in.to(at::kFloat);
      if (!cuda) {
        g = g.reshape({1});
        for (Tensor* arr : {&gp, &gt})
          if (arr->defined())
            fail(abi.scale_rows_cpu(fp(*arr), fp(g), 0, n_, at::get_num_threads()), "gd3d_scale_rows_cpu");
      } else {
        // one launch for both arrays: reads g (and the any-positive flag) on the device; leaves without touching memory when
        // g == 1 and the normal branch was taken (no host sync)
        const int32_t* flag = select_ ? (const int32_t*)(buf_.data_ptr<float>() + 1) : nullptr;
        g_finish_calls.fetch_add(1, std::memory_order_relaxed);
        fail(abi.grad_finish(fp(gp), fp(gt), fp(g), n_, flag, select_ ? fp(weight_) : nullptr,
                              (select_ && has_prologue_) ? fp(pred) : nullptr, select_ ? pro : nullptr, current_stream(pred)),
             "gd3d_grad_finish");
      }
    }
    out[0] = std::move(gp);
    out[1] = std::move(gt);
    if (twice) {   // what torch.autograd.function.once_differentiable does: an Error node behind detached aliases
      variable_list alias;
      for (auto& v : out) {
        Tensor a;
        if (v.defined()) {
          a = v.detach();
          a.set_requires_grad(true);
        }
        alias.push_back(a);
      }
      auto err = std::make_shared<torch::autograd::DelayedError>(
          "trying to differentiate twice a function that was marked with @once_differentiable", (int64_t)alias.size());
      at::AutoGradMode grad_on(true);
      return (*err)(std::move(alias));
    }
    return out;
  }
};

// One fused launch for a reduced GDLoss call.  Returns (loss sum as a 0-dim fp32 tensor, the int32 any-positive flag (1,) or
// None).  params / prologue: addresses of gd3d_params / gd3d_prologue (copied); aux: the tensor prologue->aux points into;
// ticket: device int32 of gd3d_loss_fused_one_launch or 0 (two-stage form); ev_start / ev_stop: hipEvent_t from
// gd3d_prof_event_create or 0; ws_floats: gd3d_loss_workspace_bytes(n) / 4.
std::tuple<Tensor, c10::optional<Tensor>> reduced(const Tensor& pred, const Tensor& target, const c10::optional<Tensor>& weight,
                                                  int64_t params_addr, int64_t prologue_addr, const c10::optional<Tensor>& aux,
                                                  double scale, bool select, int64_t ticket, int64_t ev_start, int64_t ev_stop,
                                                  int64_t ws_floats, bool want_flag) {
  TORCH_CHECK(pred.dim() == 2 && pred.size(1) == 7 && pred.scalar_type() == at::kFloat && pred.is_contiguous() &&
                  target.sizes() == pred.sizes() && target.scalar_type() == at::kFloat && target.is_contiguous() &&
                  target.device() == pred.device(),
              "gd3d node: pred / target must be contiguous fp32 (N, 7) tensors on one device");
  TORCH_CHECK(pred.is_cuda() || pred.is_cpu(), "gd3d node: no implementation for device ", pred.device());
  if (pred.is_cpu()) require_cpu_twins();
  TORCH_CHECK(abi.bound, "gd3d node: bind(path of libgd3d.so) has not been called");
  const int64_t n = pred.size(0);
  Tensor w;
  if (weight.has_value() && weight->defined()) {
    w = *weight;
    TORCH_CHECK(w.scalar_type() == at::kFloat && w.is_contiguous() && w.device() == pred.device() && w.size(0) == n &&
                    (w.dim() == 1 || (w.dim() == 2 && w.size(1) == 7)),
                "gd3d node: weight must be a contiguous fp32 (N,) or (N, 7) tensor on pred's device");
  }
  TORCH_CHECK(!select || (w.defined() && w.dim() == 2 && pred.is_cuda()), "gd3d node: select needs an (N, 7) weight on the GPU");
  TORCH_CHECK(prologue_addr == 0 || pred.is_cuda(), "GDLoss: the head-level fusions (bbox-coder prologue) are GPU-only");
  const bool grad_mode = at::GradMode::is_enabled();
  const bool need_gp = grad_mode && pred.requires_grad(), need_gt = grad_mode && target.requires_grad();

  std::shared_ptr<ReducedBackward> node;
  if (need_gp || need_gt) {
    node = std::shared_ptr<ReducedBackward>(new ReducedBackward(), torch::autograd::deleteNode);
    node->set_next_edges(torch::autograd::collect_next_edges(pred, target));
  }
  gd3d_params params;
  std::memcpy(&params, (const void*)params_addr, sizeof(params));
  gd3d_prologue prologue;
  const gd3d_prologue* pro = nullptr;
  if (prologue_addr != 0) {
    std::memcpy(&prologue, (const void*)prologue_addr, sizeof(prologue));
    pro = &prologue;
  }
  Tensor gp, gt, buf, total;
  {
    at::AutoDispatchBelowADInplaceOrView below_autograd;
    c10::OptionalDeviceGuard device_guard;
    if (pred.is_cuda()) device_guard.reset_device(pred.device());
    if (need_gp) gp = at::empty_like(pred);
    if (need_gt) gt = at::empty_like(target);
    // one allocation: [0] = the fp32 result, [1] = the int32 any-positive flag, [4:] = workspace (16-byte aligned)
    buf = at::empty({4 + ws_floats}, pred.options());
    float* base = buf.data_ptr<float>();
    const Weights ws(w);
    if (!pred.is_cuda()) {
      fail(abi.loss_fused_cpu(&params, fp(pred), fp(target), ws.w1, ws.w7, n, (float)scale, nullptr, base, fp(gp), fp(gt), base + 4,
                               at::get_num_threads()), "gd3d_loss_fused_cpu");
    } else {
      void* stream = current_stream(pred);
      int32_t* flag = select ? (int32_t*)(base + 1) : nullptr;
      if (ticket != 0)   // training-size call: ONE launch, the last workgroup finishes the sum
        fail(abi.loss_fused_one_launch(&params, pro, fp(pred), fp(target), ws.w1, ws.w7, n, (float)scale, base, flag, fp(gp),
                                        fp(gt), base + 4, (int32_t*)ticket, stream), "gd3d_loss_fused_one_launch");
      else if (select)
        fail(abi.loss_fused_select(&params, pro, fp(pred), fp(target), ws.w7, n, (float)scale, base, flag, fp(gp), fp(gt), base + 4,
                                    stream, (void*)ev_start, (void*)ev_stop), "gd3d_loss_fused_select");
      else
        fail(abi.loss_fused_timed(&params, pro, fp(pred), fp(target), ws.w1, ws.w7, n, (float)scale, nullptr, base, fp(gp), fp(gt),
                                   base + 4, stream, (void*)ev_start, (void*)ev_stop), "gd3d_loss_fused_timed");
    }
    total = buf.select(0, 0);
  }
  c10::optional<Tensor> flag_out;
  if (want_flag) {
    at::AutoDispatchBelowADInplaceOrView below_autograd;
    flag_out = buf.narrow(0, 1, 1).view(at::kInt);
  }
  if (node) {
    node->pred_ = torch::autograd::SavedVariable(pred, false);
    node->target_ = torch::autograd::SavedVariable(target, false);
    node->weight_ = w;
    if (aux.has_value()) node->aux_ = *aux;
    node->gp_ = std::move(gp);
    node->gt_ = std::move(gt);
    if (select) node->buf_ = buf;
    node->params_ = params;
    node->has_prologue_ = pro != nullptr;
    if (pro != nullptr) node->prologue_ = prologue;
    node->select_ = select;
    node->want_gp_ = need_gp;
    node->want_gt_ = need_gt;
    node->scale_ = (float)scale;
    node->n_ = n;
    torch::autograd::set_history(total, node);
  }
  return {total, flag_out};
}

// ---- the anchor-head regression slice (head_loss.py, SURVEY.md §8 f1: gd_anchor3d_head.py:95-161) as one node ---------------------
// Selection / gather of the positives + decode x2 + loss(es) + the gradient scattered into the NCHW head output: ONE launch
// (gd3d_anchor_head_bbox_loss[_dyn]); the zero-filled-then-scattered gradient waits in the node and backward hands it over
// (scaled by the upstream gradient on the device unless that is the library's unit gradient).  A second backward under
// retain_graph launches again.  Called from head_loss._anchor_head_fused (argument meaning documented there).
struct AnchorHeadCall {
  Tensor bbox_pred, bbox_targets, bbox_weights, anchors, sel, avg_dev;   // sel: (P,) positives or the (M,) label map (dense)
  gd3d_params params;
  gd3d_smooth_l1 sl1;
  bool has_sl1 = false, has_dw = false, dense = false, dyn = false;
  float dw[7];
  int32_t num_classes = 0;
  float scale = 0.0f;
  double w_gd = 0.0, w_sl1 = 0.0;

  // returns (loss scalar tensor, gradient | undefined)
  std::pair<Tensor, Tensor> launch(bool need_grad) const {
    const int64_t B = bbox_pred.size(0), C = bbox_pred.size(1), H = bbox_pred.size(2), W = bbox_pred.size(3);
    const int64_t P = sel.numel();
    c10::DeviceGuard device_guard(bbox_pred.device());
    Tensor grad = need_grad ? at::zeros_like(bbox_pred) : Tensor();
    Tensor buf = at::empty({4 + (int64_t)(abi.loss_workspace_bytes(P) / 4)}, bbox_pred.options());
    float* base = buf.data_ptr<float>();
    const gd3d_smooth_l1* sl = has_sl1 ? &sl1 : nullptr;
    const float* wp = bbox_weights.defined() ? bbox_weights.data_ptr<float>() : nullptr;
    const float* dwp = has_dw ? dw : nullptr;
    void* stream = current_stream(bbox_pred);
    int rc;
    if (dyn)
      rc = abi.anchor_head_dyn(&params, sl, fp(bbox_pred), (int32_t)B, (int32_t)(C / 7), (int32_t)H, (int32_t)W, fp(bbox_targets), wp, dwp,
                               fp(anchors), sel.data_ptr<int64_t>(), num_classes, w_gd, w_sl1, fp(avg_dev), base, fp(grad), base + 4,
                               stream);
    else
      rc = abi.anchor_head(&params, sl, fp(bbox_pred), (int32_t)B, (int32_t)(C / 7), (int32_t)H, (int32_t)W, fp(bbox_targets), wp, dwp,
                           fp(anchors), dense ? nullptr : sel.data_ptr<int64_t>(), P, dense ? sel.data_ptr<int64_t>() : nullptr,
                           num_classes, scale, base, fp(grad), base + 4, stream);
    fail(rc, "gd3d_anchor_head_bbox_loss");
    return {buf.select(0, 0), grad};
  }
};

struct AnchorHeadBackward : public torch::autograd::Node {
  AnchorHeadCall call_;
  torch::autograd::SavedVariable pred_;
  Tensor grad_;
  bool used_ = false;

  std::string name() const override { return "GDAnchorHeadBackward"; }
  void release_variables() override {
    std::lock_guard<std::mutex> lock(mutex_);
    pred_.reset_data();
    grad_.reset();
    call_ = AnchorHeadCall();
  }
  variable_list apply(variable_list&& grads) override {
    std::lock_guard<std::mutex> lock(mutex_);
    const bool twice = at::GradMode::is_enabled();
    at::AutoGradMode no_grad(false);
    variable_list out(1);
    const Tensor pred = pred_.unpack();   // raises after a released graph; checks in-place edits of the head output
    const Tensor& g_in = grads[0];
    if (!g_in.defined()) return out;
    Tensor g;
    if (used_) {   // retain_graph replay: the first gradient was handed over (and scaled in place): launch again
      AnchorHeadCall again = call_;
      again.bbox_pred = pred;
      g = again.launch(true).second;
    } else {
      g = std::move(grad_);
      grad_ = Tensor();
      used_ = true;
    }
    if (!is_unit_grad(g_in)) {
      const Tensor go = g_in.scalar_type() == at::kFloat ? g_in : g_in.to(at::kFloat);
      c10::DeviceGuard device_guard(g.device());
      fail(abi.scale_rows(fp(g), fp(go), 0, g.numel() / 7, current_stream(g)), "gd3d_scale_rows");
    }
    out[0] = std::move(g);
    if (twice) {
      Tensor a = out[0].detach();
      a.set_requires_grad(true);
      auto err = std::make_shared<torch::autograd::DelayedError>(
          "trying to differentiate twice a function that was marked with @once_differentiable", (int64_t)1);
      at::AutoGradMode grad_on(true);
      return (*err)(variable_list{a});
    }
    return out;
  }
};

Tensor anchor_head(const Tensor& bbox_pred, const Tensor& bbox_targets, const c10::optional<Tensor>& bbox_weights,
                   const Tensor& anchors, const Tensor& sel, int64_t params_addr, int64_t sl1_addr,
                   const c10::optional<std::vector<double>>& dw, bool dense, int64_t num_classes, double scale,
                   const c10::optional<Tensor>& avg_dev, double w_gd, double w_sl1) {
  TORCH_CHECK(abi.bound, "gd3d node: bind(path of libgd3d.so) has not been called");
  TORCH_CHECK(bbox_pred.is_cuda() && bbox_pred.dim() == 4 && bbox_pred.size(1) % 7 == 0 && bbox_pred.scalar_type() == at::kFloat &&
                  bbox_pred.is_contiguous(), "gd3d node: bbox_pred must be a contiguous fp32 (B, A*7, H, W) tensor on the GPU");
  for (const Tensor* t : {&bbox_targets, &anchors})
    TORCH_CHECK(t->scalar_type() == at::kFloat && t->is_contiguous() && t->device() == bbox_pred.device(),
                "gd3d node: targets / anchors must be contiguous fp32 tensors on bbox_pred's device");
  TORCH_CHECK(sel.scalar_type() == at::kLong && sel.is_contiguous() && sel.device() == bbox_pred.device(),
              "gd3d node: the positive list / label map must be a contiguous int64 tensor on bbox_pred's device");
  AnchorHeadCall call;
  call.bbox_pred = bbox_pred;
  call.bbox_targets = bbox_targets;
  if (bbox_weights.has_value() && bbox_weights->defined()) {
    TORCH_CHECK(bbox_weights->scalar_type() == at::kFloat && bbox_weights->is_contiguous() && bbox_weights->device() == bbox_pred.device(),
                "gd3d node: bbox_weights must be a contiguous fp32 tensor on bbox_pred's device");
    call.bbox_weights = *bbox_weights;
  }
  call.anchors = anchors;
  call.sel = sel;
  std::memcpy(&call.params, (const void*)params_addr, sizeof(call.params));
  if (sl1_addr != 0) {
    std::memcpy(&call.sl1, (const void*)sl1_addr, sizeof(call.sl1));
    call.has_sl1 = true;
  }
  if (dw.has_value()) {
    TORCH_CHECK(dw->size() == 7, "gd3d node: decode_weight must hold 7 values");
    for (int k = 0; k < 7; ++k) call.dw[k] = (float)(*dw)[k];
    call.has_dw = true;
  }
  call.dense = dense;
  call.num_classes = (int32_t)num_classes;
  call.scale = (float)scale;
  if (avg_dev.has_value() && avg_dev->defined()) {
    TORCH_CHECK(dense && avg_dev->scalar_type() == at::kFloat && avg_dev->numel() == 1 && avg_dev->device() == bbox_pred.device(),
                "gd3d node: a device-resident normaliser needs the dense form and one fp32 value on bbox_pred's device");
    call.avg_dev = *avg_dev;
    call.dyn = true;
    call.w_gd = w_gd;
    call.w_sl1 = w_sl1;
  }
  const bool need_grad = at::GradMode::is_enabled() && bbox_pred.requires_grad();
  std::shared_ptr<AnchorHeadBackward> node;
  if (need_grad) {
    node = std::shared_ptr<AnchorHeadBackward>(new AnchorHeadBackward(), torch::autograd::deleteNode);
    node->set_next_edges(torch::autograd::collect_next_edges(bbox_pred));
  }
  Tensor loss, grad;
  {
    at::AutoDispatchBelowADInplaceOrView below_autograd;
    std::tie(loss, grad) = call.launch(need_grad);
  }
  if (node) {
    node->pred_ = torch::autograd::SavedVariable(bbox_pred, false);
    call.bbox_pred = Tensor();   // the saved variable holds it (no second owner: the in-place check stays meaningful)
    node->call_ = std::move(call);
    node->grad_ = std::move(grad);
    torch::autograd::set_history(loss, node);
  }
  return loss;
}

// ---- dynamic scatter-reduce (scatter.py, SURVEY.md §8 f4: ops/voxel/scatter.py:29-72) as one node ----------------------------------
// forward: vox_scatter_reduce over the grouped points (order / seg from scatter.group_points); backward: the voxel-ordered form for
// rows of 128 bytes and more (c % 4 == 0, 32 <= c <= 256, aligned), the map-ordered gather otherwise — the choice scatter.py made.
struct ScatterReduceBackward : public torch::autograd::Node {
  // the index tensors as SavedVariables: an in-place edit of the map or of a cached grouping between forward and backward is
  // detected by version (as ctx.save_for_backward does in the Python glue), and a released graph raises torch's own message
  torch::autograd::SavedVariable pmap_, count_, argmax_, order_, seg_;
  bool has_argmax_ = false;
  int red_ = 0;
  int64_t n_ = 0, c_ = 0, v_ = 0;
  at::ScalarType in_dtype_ = at::kFloat;

  std::string name() const override { return "GDScatterReduceBackward"; }
  void release_variables() override {
    std::lock_guard<std::mutex> lock(mutex_);
    pmap_.reset_data(); count_.reset_data(); argmax_.reset_data(); order_.reset_data(); seg_.reset_data();
  }
  variable_list apply(variable_list&& grads) override {
    std::lock_guard<std::mutex> lock(mutex_);
    const bool twice = at::GradMode::is_enabled();   // create_graph=True
    at::AutoGradMode no_grad(false);
    variable_list out(1);
    const Tensor pmap = pmap_.unpack(), count = count_.unpack(), order = order_.unpack(), seg = seg_.unpack();
    const Tensor argmax = has_argmax_ ? argmax_.unpack() : Tensor();
    if (!grads[0].defined()) return out;
    const Tensor g = grads[0].contiguous().to(at::kFloat);
    c10::DeviceGuard device_guard(g.device());
    Tensor gf = at::empty({n_, c_}, g.options());
    const int32_t* am = argmax.defined() ? argmax.data_ptr<int32_t>() : nullptr;
    void* stream = current_stream(g);
    int rc;
    if (c_ % 4 == 0 && c_ >= 32 && c_ <= 256 && ((uintptr_t)g.data_ptr() & 15) == 0)
      rc = abi.scatter_backward_grouped(fp(g), order.data_ptr<int32_t>(), seg.data_ptr<int32_t>(), am, n_, (int32_t)c_, v_, red_, fp(gf),
                                        stream);
    else
      rc = abi.scatter_backward(fp(g), pmap.data_ptr<int32_t>(), count.data_ptr<int32_t>(), am, n_, (int32_t)c_, v_, red_, fp(gf), stream);
    fail(rc, "vox_scatter_backward");
    out[0] = in_dtype_ == at::kFloat ? gf : gf.to(in_dtype_);
    if (twice) {   // the gradient was written by a kernel: differentiating it again raises instead of treating it as a constant
      Tensor a = out[0].detach();
      a.set_requires_grad(true);
      auto err = std::make_shared<torch::autograd::DelayedError>(
          "trying to differentiate twice a function that was marked with @once_differentiable", (int64_t)1);
      at::AutoGradMode grad_on(true);
      return (*err)(variable_list{a});
    }
    return out;
  }
};

Tensor scatter_reduce(const Tensor& feats, const Tensor& pmap, const Tensor& count, int64_t red, const Tensor& order, const Tensor& seg) {
  TORCH_CHECK(abi.bound, "gd3d node: bind(path of libgd3d.so) has not been called");
  TORCH_CHECK(feats.is_cuda() && feats.dim() == 2, "scatter_reduce: feats must be an (N, C) tensor on the GPU");
  for (const Tensor* t : {&pmap, &count, &order, &seg})
    TORCH_CHECK(t->scalar_type() == at::kInt && t->is_contiguous() && t->device() == feats.device(),
                "scatter_reduce: map / count / order / seg must be contiguous int32 tensors on feats' device");
  const int64_t n = feats.size(0), c = feats.size(1), v = count.numel();
  TORCH_CHECK(pmap.numel() == n && order.numel() == n && seg.numel() == v + 1 && red >= 0 && red <= 2,
              "scatter_reduce: inconsistent index tensors");
  const bool need_grad = at::GradMode::is_enabled() && feats.requires_grad();
  std::shared_ptr<ScatterReduceBackward> node;
  if (need_grad) {
    node = std::shared_ptr<ScatterReduceBackward>(new ScatterReduceBackward(), torch::autograd::deleteNode);
    node->set_next_edges(torch::autograd::collect_next_edges(feats));
  }
  Tensor out, argmax, result;
  {
    at::AutoDispatchBelowADInplaceOrView below_autograd;
    c10::DeviceGuard device_guard(feats.device());
    const Tensor f32 = feats.scalar_type() == at::kFloat ? feats.contiguous() : feats.to(at::kFloat).contiguous();
    out = at::empty({v, c}, f32.options());
    if (red == 2) argmax = at::empty({v, c}, f32.options().dtype(at::kInt));
    fail(abi.scatter_reduce(fp(f32), order.data_ptr<int32_t>(), seg.data_ptr<int32_t>(), n, (int32_t)c, v, (int)red, fp(out),
                            argmax.defined() ? argmax.data_ptr<int32_t>() : nullptr, current_stream(feats)), "vox_scatter_reduce");
    result = feats.scalar_type() == at::kFloat ? out : out.to(feats.scalar_type());
  }
  if (node) {
    using torch::autograd::SavedVariable;
    node->pmap_ = SavedVariable(pmap, false); node->count_ = SavedVariable(count, false);
    node->order_ = SavedVariable(order, false); node->seg_ = SavedVariable(seg, false);
    node->has_argmax_ = argmax.defined();
    if (argmax.defined()) node->argmax_ = SavedVariable(argmax, false);
    node->red_ = (int)red; node->n_ = n; node->c_ = c; node->v_ = v; node->in_dtype_ = feats.scalar_type();
    torch::autograd::set_history(result, node);
  }
  return result;
}

// nms_gpu's scored path (mmdet3d-gaussian_amd/iou3d.py: <= rnms_scored_max_n() candidates, fp32 scores): the three allocations, the
// launch and — unless `padded` — the one read-back of the count and the cut to it, without the Python in between (a third of
// nms_gpu's end-to-end time at inference sizes was host code).  boxes (N,5) / scores (N) contiguous fp32 on one GPU.
// Returns (keep, num): padded: keep (n_keep) int64 whose first num[0] entries are valid, num (1) int64 on the device;
// otherwise keep is already cut to the count (and to post_max when >= 0) and num is undefined.
std::tuple<Tensor, Tensor> nms_scored(const Tensor& boxes, const Tensor& scores, double thresh, int64_t n_keep, bool normal,
                                      bool padded, int64_t post_max) {
  TORCH_CHECK(abi.bound, "gd3d node: bind(path of libgd3d.so) has not been called");
  TORCH_CHECK(boxes.is_cuda() && boxes.dim() == 2 && boxes.size(1) == 5 && boxes.scalar_type() == at::kFloat && boxes.is_contiguous() &&
                  scores.dim() == 1 && scores.size(0) == boxes.size(0) && scores.scalar_type() == at::kFloat &&
                  scores.is_contiguous() && scores.device() == boxes.device() && n_keep > 0 && n_keep <= boxes.size(0),
              "gd3d node: nms_scored takes contiguous fp32 (N,5) boxes and (N) scores on one GPU and 0 < n_keep <= N");
  const int64_t n_all = boxes.size(0);
  c10::DeviceGuard device_guard(boxes.device());
  const auto longs = boxes.options().dtype(at::kLong);
  Tensor keep = at::empty({n_keep}, longs), num;
  Tensor ws = at::empty({(int64_t)abi.nms_scored_workspace_bytes(n_all, n_keep)}, boxes.options().dtype(at::kByte));
  // The one unavoidable wait: the result length is data dependent.  Unpadded calls let the scan kernel write the count straight
  // into PINNED HOST memory (a thread's own mailbox word) and poll it — no copy call, no stream synchronisation: 55.0 -> 49.8 us
  // per call at n = 4096 against the blocking 8-byte copy (profiles/r06_nms_batched.txt).  The kept ids stay on the device and
  // are stream-ordered as before; if the word does not arrive within 0.2 s the stream is synchronised and the word read once more.
  static thread_local Tensor mailbox;
  volatile int64_t* word = nullptr;
  constexpr int64_t PENDING = -(int64_t(1) << 62);
  if (padded) {
    num = at::empty({1}, longs);
  } else {
    if (!mailbox.defined()) mailbox = at::empty({8}, at::TensorOptions().dtype(at::kLong).pinned_memory(true));
    word = mailbox.data_ptr<int64_t>();
    *word = PENDING;
  }
  fail(abi.nms_scored(normal ? 1 : 0, boxes.data_ptr<float>(), scores.data_ptr<float>(), n_all, n_keep, (float)thresh,
                      keep.data_ptr<int64_t>(), padded ? num.data_ptr<int64_t>() : (int64_t*)word, ws.data_ptr(), current_stream(boxes)),
       normal ? "nms_normal_gpu" : "nms_gpu");
  if (padded) return {keep, num};
  int64_t k = PENDING;
  {
    pybind11::gil_scoped_release nogil;
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0; (k = *word) == PENDING; ++spins) {
      if ((spins & 0xfff) == 0xfff && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200)) {
        c10::hip::getCurrentHIPStream(boxes.device().index()).synchronize();
        k = *word;
        break;
      }
    }
  }
  TORCH_CHECK(k != PENDING, "nms_gpu: the NMS kernels finished without reporting a count");
  TORCH_CHECK(k >= 0, "nms_gpu: the device-side NMS scan gave up (num_keep = ", k, "); the result is void");   // failure mark of the list scan
  if (post_max >= 0 && k > post_max) k = post_max;
  return {keep.narrow(0, 0, k), Tensor()};
}

void set_unit_grad(int64_t device_index, int64_t address) {
  TORCH_CHECK(device_index < MAX_DEVICES, "gd3d node: device index ", device_index);
  const int slot = device_index < 0 ? MAX_DEVICES : (int)device_index;   // < 0: the CPU's constant
  g_unit_grad[slot] = (const void*)address;
}

}  // namespace

PYBIND11_MODULE(_gd3d_node, m) {
  m.doc() = "C++ autograd node of GDLoss's reduced forms above the C ABI of libgd3d.so";
  m.def("reduced", &reduced, py::arg("pred"), py::arg("target"), py::arg("weight"), py::arg("params"), py::arg("prologue"),
        py::arg("aux"), py::arg("scale"), py::arg("select"), py::arg("ticket"), py::arg("ev_start"), py::arg("ev_stop"),
        py::arg("ws_floats"), py::arg("want_flag"));
  m.def("nms_scored", &nms_scored, py::arg("boxes"), py::arg("scores"), py::arg("thresh"), py::arg("n_keep"), py::arg("normal"),
        py::arg("padded"), py::arg("post_max"));
  m.def("anchor_head", &anchor_head, py::arg("bbox_pred"), py::arg("bbox_targets"), py::arg("bbox_weights"), py::arg("anchors"),
        py::arg("sel"), py::arg("params"), py::arg("sl1"), py::arg("decode_weight"), py::arg("dense"), py::arg("num_classes"),
        py::arg("scale"), py::arg("avg_dev"), py::arg("w_gd"), py::arg("w_sl1"));
  m.def("scatter_reduce", &scatter_reduce, py::arg("feats"), py::arg("point2voxel_map"), py::arg("voxel_points_count"), py::arg("reduce"),
        py::arg("order"), py::arg("seg"));
  m.def("set_unit_grad", &set_unit_grad);
  m.def("finish_calls", []() { return g_finish_calls.load(); }, "gd3d_grad_finish launches made by backward so far");
  m.def("bind", &bind, "resolve the C ABI from the loaded libgd3d.so; returns its ABI version");
}
