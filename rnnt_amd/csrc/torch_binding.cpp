// torch_binding.cpp — the compiled PyTorch binding of librnnt_mi355.so (rnnt_amd/lib/librnnt_amd_torch.so).
//
// What upstream's `warprnnt_pytorch` extension is to `warp-transducer` (the operator the reference calls at
// /root/reference/src/rnnt/model.py:5,87,116): a thin C++ layer between torch tensors and the C ABI of
// include/rnnt_mi355.h.  It takes the launch stream from torch's current HIP stream and every buffer (costs,
// loss, lattice workspace, gradients) from torch's caching allocator, calls rnnt_forward / rnnt_backward, and
// records the backward pass as a C++ autograd node, so a training step crosses the Python/C boundary once.
// No kernels here and no arithmetic: the C ABI underneath is the same one the ctypes binding
// (rnnt_amd/_lib.py) drives.
//
//   torch.ops.rnnt_amd.rnnt_loss(acts, labels, act_lens, label_lens, blank, reduction, fwd_events, bwd_events) -> loss
//   torch.ops.rnnt_amd.rnnt_forward(acts, labels, act_lens, label_lens, blank, reduction) -> (loss, costs, workspace)
//   torch.ops.rnnt_amd.rnnt_backward(acts, labels, act_lens, label_lens, grad_out, workspace, blank, reduction) -> grads
//
// reduction: 0 none, 1 mean, 2 sum (rnntReduction_t).  fwd_events / bwd_events: address of an array of three
// hipEvent handles the library records at its stage boundaries (bench.py), or 0.
#include <ATen/ATen.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <torch/csrc/autograd/custom_function.h>
#include <torch/library.h>

#include <tuple>

#include "../../include/rnnt_mi355.h"

namespace {

void check_status(rnntStatus_t s, const char *what) {
    TORCH_CHECK(s == RNNT_STATUS_SUCCESS, what, ": ", rnntGetStatusString(s), " (rnntStatus_t=", (int)s, ")");
}

int dtype_code(const at::Tensor &acts) {
    switch (acts.scalar_type()) {
        case at::kFloat: return RNNT_DTYPE_F32;
        case at::kBFloat16: return RNNT_DTYPE_BF16;
        case at::kHalf: return RNNT_DTYPE_F16;
        default: TORCH_CHECK_TYPE(false, "acts must be torch.float32, torch.bfloat16 or torch.float16");
    }
}

// The Python operator (rnnt_amd/loss.py: certify_inputs) has checked ranks, dtypes and shapes with upstream's
// messages; these are the conditions the C ABI itself relies on, for callers of torch.ops directly.
void check_inputs(const at::Tensor &acts, const at::Tensor &labels, const at::Tensor &act_lens,
                  const at::Tensor &label_lens) {
    TORCH_CHECK(acts.is_cuda(), "rnnt_amd computes on MI355X (ROCm device tensors) only; got acts on ", acts.device(),
                ". There is no CPU path or PyTorch fallback in this package.");
    TORCH_CHECK(labels.device() == acts.device() && act_lens.device() == acts.device() &&
                    label_lens.device() == acts.device(),
                "labels, lengths and label_lengths must be on the same device as acts");
    TORCH_CHECK_VALUE(acts.dim() == 4 && labels.dim() == 2 && act_lens.dim() == 1 && label_lens.dim() == 1,
                      "acts must be 4D, labels 2D, lengths and label_lengths 1D");
    TORCH_CHECK_TYPE(labels.scalar_type() == at::kInt && act_lens.scalar_type() == at::kInt &&
                         label_lens.scalar_type() == at::kInt,
                     "labels, lengths and label_lengths must be torch.int32");
    TORCH_CHECK_VALUE(acts.is_contiguous() && labels.is_contiguous() && act_lens.is_contiguous() &&
                          label_lens.is_contiguous(),
                      "acts, labels, lengths and label_lengths must be contiguous");
    TORCH_CHECK_VALUE(acts.size(0) >= 1 && acts.size(1) >= 1 && acts.size(2) >= 1 && acts.size(3) >= 1,
                      "acts must not be empty");
    TORCH_CHECK_VALUE(labels.size(0) == acts.size(0) && act_lens.size(0) == acts.size(0) &&
                          label_lens.size(0) == acts.size(0) && labels.size(1) == acts.size(2) - 1,
                      "labels must be [B, U] and the length tensors [B] for acts [B, T, U+1, V]");
}

rnntOptions options_for(const at::Tensor &acts, int64_t blank, int64_t events) {
    rnntOptions o;
    o.stream = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(acts.get_device()).stream();
    o.blank_label = (int)blank;
    o.maxT = (int)acts.size(1);
    o.maxU = (int)acts.size(2);
    o.dtype = dtype_code(acts);
    o.stage_events = reinterpret_cast<void **>(static_cast<intptr_t>(events));
    return o;
}

std::tuple<at::Tensor, at::Tensor, at::Tensor> forward_impl(const at::Tensor &acts, const at::Tensor &labels,
                                                           const at::Tensor &act_lens, const at::Tensor &label_lens,
                                                           int64_t blank, int64_t reduction, int64_t events) {
    check_inputs(acts, labels, act_lens, label_lens);
    TORCH_CHECK_VALUE(reduction >= RNNT_REDUCTION_NONE && reduction <= RNNT_REDUCTION_SUM,
                      "reduction must be 0 (none), 1 (mean) or 2 (sum)");
    const c10::hip::HIPGuardMasqueradingAsCUDA guard(acts.device());
    const int B = (int)acts.size(0), V = (int)acts.size(3);
    const rnntOptions o = options_for(acts, blank, events);
    size_t ws_bytes = 0;
    check_status(rnnt_get_workspace_size(o.maxT, o.maxU, B, &ws_bytes), "rnnt_get_workspace_size");
    const auto f32 = acts.options().dtype(at::kFloat);
    at::Tensor ws = at::empty({(int64_t)ws_bytes}, acts.options().dtype(at::kByte));  // caching allocator: 512-byte aligned
    at::Tensor costs = at::empty({B}, f32);
    at::Tensor loss = at::empty({reduction == RNNT_REDUCTION_NONE ? B : 1}, f32);
    check_status(rnnt_forward(acts.data_ptr(), labels.data_ptr<int>(), act_lens.data_ptr<int>(),
                              label_lens.data_ptr<int>(), V, B, costs.data_ptr<float>(), loss.data_ptr<float>(),
                              (int)reduction, ws.data_ptr(), o),
                 "rnnt_forward");
    return {loss, costs, ws};
}

at::Tensor backward_impl(const at::Tensor &acts, const at::Tensor &labels, const at::Tensor &act_lens,
                         const at::Tensor &label_lens, const at::Tensor &grad_out, const at::Tensor &workspace,
                         int64_t blank, int64_t reduction, int64_t events) {
    check_inputs(acts, labels, act_lens, label_lens);
    const c10::hip::HIPGuardMasqueradingAsCUDA guard(acts.device());
    const int B = (int)acts.size(0), V = (int)acts.size(3);
    const rnntOptions o = options_for(acts, blank, events);
    size_t ws_bytes = 0;
    check_status(rnnt_get_workspace_size(o.maxT, o.maxU, B, &ws_bytes), "rnnt_get_workspace_size");
    TORCH_CHECK(workspace.is_cuda() && workspace.device() == acts.device() && workspace.is_contiguous() &&
                    (size_t)workspace.nbytes() >= ws_bytes,
                "workspace must be the one rnnt_forward filled for these inputs");
    at::Tensor go = grad_out.to(acts.device(), at::kFloat).contiguous().reshape({-1});
    int stride = 0;
    float scale = 1.0f;
    if (reduction == RNNT_REDUCTION_NONE) {
        TORCH_CHECK(go.numel() == B, "grad_output must have one element per utterance");
        stride = 1;
    } else {
        TORCH_CHECK(go.numel() >= 1, "grad_output must not be empty");
        scale = reduction == RNNT_REDUCTION_MEAN ? 1.0f / (float)B : 1.0f;
    }
    at::Tensor grads = at::empty_like(acts);
    check_status(rnnt_backward(acts.data_ptr(), grads.data_ptr(), labels.data_ptr<int>(), act_lens.data_ptr<int>(),
                               label_lens.data_ptr<int>(), V, B, go.data_ptr<float>(), stride, scale,
                               workspace.data_ptr(), o),
                 "rnnt_backward");
    return grads;
}

class RnntLossFunction : public torch::autograd::Function<RnntLossFunction> {
public:
    static at::Tensor forward(torch::autograd::AutogradContext *ctx, const at::Tensor &acts, const at::Tensor &labels,
                              const at::Tensor &act_lens, const at::Tensor &label_lens, int64_t blank,
                              int64_t reduction, int64_t fwd_events, int64_t bwd_events) {
        auto out = forward_impl(acts, labels, act_lens, label_lens, blank, reduction, fwd_events);
        ctx->save_for_backward({acts, labels, act_lens, label_lens, std::get<2>(out)});
        ctx->saved_data["blank"] = blank;
        ctx->saved_data["reduction"] = reduction;
        ctx->saved_data["bwd_events"] = bwd_events;
        return std::get<0>(out);
    }

    static torch::autograd::variable_list backward(torch::autograd::AutogradContext *ctx,
                                                   torch::autograd::variable_list grad_outputs) {
        const auto saved = ctx->get_saved_variables();
        at::Tensor grads = backward_impl(saved[0], saved[1], saved[2], saved[3], grad_outputs[0], saved[4],
                                         ctx->saved_data["blank"].toInt(), ctx->saved_data["reduction"].toInt(),
                                         ctx->saved_data["bwd_events"].toInt());
        return {grads, at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor()};
    }
};

at::Tensor rnnt_loss_op(const at::Tensor &acts, const at::Tensor &labels, const at::Tensor &act_lens,
                        const at::Tensor &label_lens, int64_t blank, int64_t reduction, int64_t fwd_events,
                        int64_t bwd_events) {
    return RnntLossFunction::apply(acts, labels, act_lens, label_lens, blank, reduction, fwd_events, bwd_events);
}

std::tuple<at::Tensor, at::Tensor, at::Tensor> rnnt_forward_op(const at::Tensor &acts, const at::Tensor &labels,
                                                              const at::Tensor &act_lens,
                                                              const at::Tensor &label_lens, int64_t blank,
                                                              int64_t reduction) {
    return forward_impl(acts, labels, act_lens, label_lens, blank, reduction, 0);
}

at::Tensor rnnt_backward_op(const at::Tensor &acts, const at::Tensor &labels, const at::Tensor &act_lens,
                            const at::Tensor &label_lens, const at::Tensor &grad_out, const at::Tensor &workspace,
                            int64_t blank, int64_t reduction) {
    return backward_impl(acts, labels, act_lens, label_lens, grad_out, workspace, blank, reduction, 0);
}

int64_t abi_version_op() { return rnnt_mi355_abi_version(); }

}  // namespace

TORCH_LIBRARY(rnnt_amd, m) {
    m.def("rnnt_loss(Tensor acts, Tensor labels, Tensor act_lens, Tensor label_lens, int blank, int reduction, "
          "int fwd_events=0, int bwd_events=0) -> Tensor",
          &rnnt_loss_op);
    m.def("rnnt_forward(Tensor acts, Tensor labels, Tensor act_lens, Tensor label_lens, int blank, int reduction) "
          "-> (Tensor, Tensor, Tensor)",
          &rnnt_forward_op);
    m.def("rnnt_backward(Tensor acts, Tensor labels, Tensor act_lens, Tensor label_lens, Tensor grad_out, "
          "Tensor workspace, int blank, int reduction) -> Tensor",
          &rnnt_backward_op);
    m.def("abi_version() -> int", &abi_version_op);
}
