// TORCH_LIBRARY(ssdhip): the torch-operator face of libssdhip.so ("a PyTorch-ROCm C++/HIP extension registering
// TORCH_LIBRARY(ssdhip) ops, each a thin wrapper over an extern "C" launcher").  No kernel lives here: every op validates its
// tensors, borrows their device pointers and the current HIP stream of their device, calls ONE entry point of include/ssdhip.h and
// turns a negative return code into a c10::Error carrying ssdhip_last_error() (no exception crosses the C ABI).  Host-only C++
// (g++ against the torch headers); built by `make torch_ops` into lib/libssdhip_torch.so, loaded with torch.ops.load_library.
//
// Reference call sites replaced (paths relative to the reference's src/): see include/ssdhip.h per entry point.
#include <ATen/ATen.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <torch/library.h>

#include <vector>

#include "../../include/ssdhip.h"

namespace {

void check(int rc, const char* what) {
    if (rc >= 0) return;
    const char* msg = ssdhip_last_error();
    // the Python mirror raises ValueError for bad arguments like the reference's torch calls do; c10::ValueError maps to it
    if (rc == SSDHIP_EINVAL) { TORCH_CHECK_VALUE(false, what, ": ", msg ? msg : "invalid argument"); }
    TORCH_CHECK(false, what, ": ", msg ? msg : "launch failed", " (code ", rc, ")");
}

const float* fptr(const at::Tensor& t) { return t.defined() ? t.data_ptr<float>() : nullptr; }
const float* fptr(const c10::optional<at::Tensor>& t) { return (t.has_value() && t->defined()) ? t->data_ptr<float>() : nullptr; }
float* fptr_mut(const c10::optional<at::Tensor>& t) { return (t.has_value() && t->defined()) ? t->data_ptr<float>() : nullptr; }

void need(const at::Tensor& t, const char* name, at::ScalarType dt = at::kFloat) {
    TORCH_CHECK(t.defined() && t.is_cuda(), "ssdhip: ", name, " must live on the GPU (there is no CPU fallback)");
    TORCH_CHECK_VALUE(t.scalar_type() == dt && t.is_contiguous(), "ssdhip: ", name, " must be contiguous ", c10::toString(dt));
}
void need_opt(const c10::optional<at::Tensor>& t, const char* name) { if (t.has_value() && t->defined()) need(*t, name); }

struct Ctx {       // device guard + the stream torch is enqueueing on for that device (ROCm builds of torch name the device type "cuda")
    c10::hip::HIPGuardMasqueradingAsCUDA guard;
    void* stream;
    explicit Ctx(const at::Tensor& t)
        : guard(t.device()), stream((void*)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.device().index()).stream()) {}
};

// ---- A9 SSD._get_default_bboxes (model/ssd.py:108-133): maps = flat [m, n, anchors] per feature map, host ints
at::Tensor default_boxes(at::IntArrayRef maps, double s_min, double s_max, at::Device device) {
    TORCH_CHECK_VALUE(maps.size() % 3 == 0 && !maps.empty(), "ssdhip::default_boxes: maps must be a flat list of (m, n, anchors) triples");
    std::vector<int32_t> host(maps.begin(), maps.end());
    int64_t P = 0;
    for (size_t k = 0; k < host.size(); k += 3) P += (int64_t)host[k] * host[k + 1] * host[k + 2];
    at::Tensor out = at::empty({P, 4}, at::TensorOptions().device(device).dtype(at::kFloat));
    Ctx c(out);
    check(ssdhip_default_boxes(out.data_ptr<float>(), host.data(), (int)(host.size() / 3), s_min, s_max, c.stream), "ssdhip::default_boxes");
    return out;
}

// ---- A2 VGG16.normalize fused with NCHW -> NHWC4 (model/vgg16.py:103-115)
at::Tensor normalize_nchw_to_nhwc4(const at::Tensor& x) {
    need(x, "x");
    TORCH_CHECK_VALUE(x.dim() == 4 && x.size(1) == 3, "ssdhip::normalize_nchw_to_nhwc4: expected (N, 3, H, W)");
    at::Tensor y = at::empty({x.size(0), x.size(2), x.size(3), 4}, x.options());
    Ctx c(x);
    check(ssdhip_normalize_nchw_to_nhwc4(x.data_ptr<float>(), y.data_ptr<float>(), (int)x.size(0), (int)x.size(2), (int)x.size(3), c.stream),
          "ssdhip::normalize_nchw_to_nhwc4");
    return y;
}

// ---- weights: OIHW -> packed implicit-GEMM operand (mode 0 forward, 1 data gradient)
at::Tensor pack_conv_weight(const at::Tensor& w, int64_t rows_pad, int64_t inner_pad, int64_t KP, int64_t mode) {
    need(w, "w");
    TORCH_CHECK_VALUE(w.dim() == 4, "ssdhip::pack_conv_weight: OIHW weights");
    at::Tensor packed = at::empty({rows_pad * KP}, w.options());
    Ctx c(w);
    check(ssdhip_pack_conv_weight(w.data_ptr<float>(), packed.data_ptr<float>(), (int)w.size(0), (int)w.size(1), (int)w.size(2), (int)w.size(3),
                                  (int)rows_pad, (int)inner_pad, (int)KP, (int)mode, c.stream), "ssdhip::pack_conv_weight");
    return packed;
}

// ---- A3/A7 nn.Conv2d forward (mode 0) / data gradient (mode 1) as NHWC implicit GEMM (model/vgg16.py:33, model/ssd.py:61,70-77)
// x (N,Hi,Wi,Cin) NHWC, packed weights from pack_conv_weight; returns y (N,Ho,Wo,Cout) and, with want_stats, the BatchNorm partial table.
std::tuple<at::Tensor, at::Tensor> conv2d_nhwc(const at::Tensor& x, const at::Tensor& w_packed, const c10::optional<at::Tensor>& bias,
                                               const c10::optional<at::Tensor>& in_scale, const c10::optional<at::Tensor>& in_shift,
                                               int64_t Cout, int64_t k, int64_t stride, int64_t pad, int64_t KP, int64_t mode,
                                               at::IntArrayRef out_hw, bool want_stats, bool bf16) {
    need(x, "x"); need(w_packed, "w_packed"); need_opt(bias, "bias"); need_opt(in_scale, "in_scale"); need_opt(in_shift, "in_shift");
    TORCH_CHECK_VALUE(x.dim() == 4 && out_hw.size() == 2, "ssdhip::conv2d_nhwc: x must be (N, H, W, C), out_hw = (Ho, Wo)");
    ssdhip_conv_params p{};
    p.x = x.data_ptr<float>(); p.w = w_packed.data_ptr<float>(); p.bias = fptr(bias); p.in_scale = fptr(in_scale); p.in_shift = fptr(in_shift);
    p.N = (int)x.size(0); p.Hi = (int)x.size(1); p.Wi = (int)x.size(2); p.Cin = (int)x.size(3);
    p.Ho = (int)out_hw[0]; p.Wo = (int)out_hw[1]; p.Cout = (int)Cout;
    p.R = p.S = (int)k; p.stride = (int)stride; p.pad = (int)pad; p.KP = (int)KP; p.mode = (int)mode; p.compute = bf16 ? 1 : 0;
    at::Tensor y = at::empty({x.size(0), out_hw[0], out_hw[1], Cout}, x.options());
    p.y = y.data_ptr<float>();
    p.y_batch_stride = out_hw[0] * out_hw[1] * Cout;
    at::Tensor stats;
    if (want_stats) {
        const int bm = ssdhip_conv2d_block_m(&p);
        check(bm, "ssdhip::conv2d_nhwc (tile choice)");
        const int64_t M = (int64_t)p.N * p.Ho * p.Wo;
        stats = at::empty({(M + bm - 1) / bm, Cout, 2}, x.options());
        p.stats = stats.data_ptr<float>();
    }
    at::Tensor ws;
    const int64_t need_ws = ssdhip_conv2d_ws_bytes(&p);
    if (need_ws > 0) {
        ws = at::empty({(need_ws + 3) / 4}, x.options());
        p.ws = ws.data_ptr<float>(); p.ws_bytes = ws.numel() * 4;
    }
    Ctx c(x);
    check(ssdhip_conv2d_nhwc(&p, c.stream), "ssdhip::conv2d_nhwc");
    return {y, stats};
}

// ---- autograd backward of nn.Conv2d: weight (+ bias) gradient (train.py:121)
std::tuple<at::Tensor, at::Tensor> conv2d_wgrad(const at::Tensor& x, const at::Tensor& dy, at::IntArrayRef w_shape, int64_t stride, int64_t pad,
                                                const c10::optional<at::Tensor>& in_scale, const c10::optional<at::Tensor>& in_shift,
                                                bool want_bias, int64_t winograd, bool bf16) {
    need(x, "x"); need(dy, "dy"); need_opt(in_scale, "in_scale"); need_opt(in_shift, "in_shift");
    TORCH_CHECK_VALUE(x.dim() == 4 && dy.dim() == 4 && w_shape.size() == 4, "ssdhip::conv2d_wgrad: x, dy NHWC and an OIHW weight shape");
    TORCH_CHECK_VALUE(winograd == 0 || winograd == 2 || winograd == 4, "ssdhip::conv2d_wgrad: winograd must be 0 (direct), 2 (F(2x2)) or 4 (F(4x4))");
    ssdhip_wgrad_params p{};
    at::Tensor dw = at::empty(w_shape, x.options());
    at::Tensor db = want_bias ? at::empty({w_shape[0]}, x.options()) : at::Tensor();
    p.x = x.data_ptr<float>(); p.dy = dy.data_ptr<float>(); p.in_scale = fptr(in_scale); p.in_shift = fptr(in_shift);
    p.dw_oihw = dw.data_ptr<float>(); p.dbias = want_bias ? db.data_ptr<float>() : nullptr;
    p.N = (int)x.size(0); p.Hi = (int)x.size(1); p.Wi = (int)x.size(2); p.Cin = (int)w_shape[1];
    p.Ho = (int)dy.size(1); p.Wo = (int)dy.size(2); p.Cout = (int)w_shape[0]; p.R = p.S = (int)w_shape[2];
    p.stride = (int)stride; p.pad = (int)pad; p.dy_row = (int)dy.size(3); p.dy_batch_stride = dy.size(1) * dy.size(2) * dy.size(3);
    p.compute = bf16 ? 1 : 0;
    const int64_t bytes = winograd == 4 ? ssdhip_conv2d_wgrad_winograd4_ws_bytes(&p) : winograd == 2 ? ssdhip_conv2d_wgrad_winograd_ws_bytes(&p)
                                                                                                       : ssdhip_conv2d_wgrad_ws_bytes(&p);
    check((int)std::min<int64_t>(bytes, 0), "ssdhip::conv2d_wgrad (workspace query)");
    at::Tensor ws = at::empty({(bytes + 3) / 4 + 1}, x.options());
    p.ws = ws.data_ptr<float>();
    Ctx c(x);
    const int rc = winograd == 4 ? ssdhip_conv2d_wgrad_winograd4(&p, c.stream) : winograd == 2 ? ssdhip_conv2d_wgrad_winograd(&p, c.stream)
                                                                                                : ssdhip_conv2d_wgrad(&p, c.stream);
    check(rc, "ssdhip::conv2d_wgrad");
    return {dw, db};
}

// ---- A4 nn.BatchNorm2d(train): partial table -> (scale, shift, mean, invstd); running stats updated in place like the module
std::tuple<at::Tensor, at::Tensor, at::Tensor, at::Tensor> bn_finalize(const at::Tensor& stats, int64_t count, const at::Tensor& gamma, const at::Tensor& beta,
                                                                       c10::optional<at::Tensor> running_mean, c10::optional<at::Tensor> running_var,
                                                                       double momentum, double eps) {
    need(stats, "stats"); need(gamma, "gamma"); need(beta, "beta"); need_opt(running_mean, "running_mean"); need_opt(running_var, "running_var");
    TORCH_CHECK_VALUE(stats.dim() == 3 && stats.size(2) == 2, "ssdhip::bn_finalize: stats must be (rows, C, 2)");
    const int64_t C = stats.size(1);
    at::Tensor scale = at::empty({C}, stats.options()), shift = at::empty({C}, stats.options()), mean = at::empty({C}, stats.options()),
               invstd = at::empty({C}, stats.options());
    at::Tensor ws = at::empty({64 * C * 2}, stats.options().dtype(at::kDouble));
    Ctx c(stats);
    check(ssdhip_bn_finalize(stats.data_ptr<float>(), (int)stats.size(0), (int)C, count, gamma.data_ptr<float>(), beta.data_ptr<float>(),
                             fptr_mut(running_mean), fptr_mut(running_var), (float)momentum, (float)eps, scale.data_ptr<float>(), shift.data_ptr<float>(),
                             mean.data_ptr<float>(), invstd.data_ptr<float>(), ws.data_ptr<double>(), c.stream), "ssdhip::bn_finalize");
    return {scale, shift, mean, invstd};
}

// ---- A4-A6 BN apply + ReLU (+ MaxPool2d(2, 2, pad)) (model/vgg16.py:28-35)
at::Tensor bn_relu_pool(const at::Tensor& z, const at::Tensor& scale, const at::Tensor& shift, bool pool, int64_t pool_pad) {
    need(z, "z"); need(scale, "scale"); need(shift, "shift");
    TORCH_CHECK_VALUE(z.dim() == 4, "ssdhip::bn_relu_pool: z must be (N, H, W, C)");
    const int64_t H = z.size(1), W = z.size(2);
    const int64_t ho = pool ? (H + 2 * pool_pad - 2) / 2 + 1 : H, wo = pool ? (W + 2 * pool_pad - 2) / 2 + 1 : W;
    at::Tensor y = at::empty({z.size(0), ho, wo, z.size(3)}, z.options());
    Ctx c(z);
    check(ssdhip_bn_relu_pool(z.data_ptr<float>(), scale.data_ptr<float>(), shift.data_ptr<float>(), y.data_ptr<float>(), (int)z.size(0), (int)H, (int)W,
                              (int)z.size(3), pool ? 1 : 0, (int)pool_pad, c.stream), "ssdhip::bn_relu_pool");
    return y;
}

// ---- A10-A16 SSD.loss (model/ssd.py:181-328): returns (loss (1), terms (3,N,P), sel (N,8)); terms / sel feed the backward
std::tuple<at::Tensor, at::Tensor, at::Tensor> multibox_loss_fwd(const at::Tensor& outputs, const at::Tensor& targets, const at::Tensor& dboxes,
                                                                 double alpha, double threshold) {
    need(outputs, "outputs"); need(targets, "targets"); need(dboxes, "default_bboxes");
    TORCH_CHECK_VALUE(outputs.dim() == 3 && targets.dim() == 3 && outputs.size(0) == targets.size(0) && outputs.size(2) == targets.size(2),
                      "ssdhip::multibox_loss_fwd: outputs (N, P, 4 + C) and targets (N, G, 4 + C) do not match");
    TORCH_CHECK_VALUE(dboxes.dim() == 2 && dboxes.size(0) == outputs.size(1) && dboxes.size(1) == 4, "ssdhip::multibox_loss_fwd: default_bboxes must be (P, 4)");
    const int64_t N = outputs.size(0), P = outputs.size(1), G = targets.size(1), row = outputs.size(2);
    at::Tensor terms = at::empty({3, N, P}, outputs.options()), sel = at::empty({N, 8}, outputs.options()), loss = at::empty({1}, outputs.options());
    at::Tensor cnt = at::empty({N}, outputs.options().dtype(at::kInt));
    Ctx c(outputs);
    check(ssdhip_multibox_loss_fwd(outputs.data_ptr<float>(), targets.data_ptr<float>(), dboxes.data_ptr<float>(), (int)N, (int)P, (int)G, (int)(row - 4),
                                   (float)alpha, (float)threshold, terms.data_ptr<float>(), cnt.data_ptr<int32_t>(), sel.data_ptr<float>(),
                                   loss.data_ptr<float>(), c.stream), "ssdhip::multibox_loss_fwd");
    return {loss, terms, sel};
}

at::Tensor multibox_loss_bwd(const at::Tensor& outputs, const at::Tensor& targets, const at::Tensor& dboxes, double alpha, double threshold,
                             const at::Tensor& terms, const at::Tensor& sel, const at::Tensor& grad_loss) {
    need(outputs, "outputs"); need(targets, "targets"); need(dboxes, "default_bboxes"); need(terms, "terms"); need(sel, "sel"); need(grad_loss, "grad_loss");
    TORCH_CHECK_VALUE(grad_loss.numel() == 1, "ssdhip::multibox_loss_bwd: grad_loss must hold one value");
    const int64_t N = outputs.size(0), P = outputs.size(1), G = targets.size(1), row = outputs.size(2);
    at::Tensor grad = at::empty_like(outputs);
    Ctx c(outputs);
    check(ssdhip_multibox_loss_bwd(outputs.data_ptr<float>(), targets.data_ptr<float>(), dboxes.data_ptr<float>(), (int)N, (int)P, (int)G, (int)(row - 4),
                                   (float)alpha, (float)threshold, terms.data_ptr<float>(), sel.data_ptr<float>(), grad_loss.data_ptr<float>(),
                                   grad.data_ptr<float>(), c.stream), "ssdhip::multibox_loss_bwd");
    return grad;
}

// ---- A19 + A20 utils.calc_coordicate + utils.calc_score (utils.py:19-55), in place
void decode_score_(at::Tensor outputs, const at::Tensor& dboxes) {
    need(outputs, "outputs"); need(dboxes, "default_bboxes");
    TORCH_CHECK_VALUE(outputs.dim() == 3 && outputs.size(2) >= 5, "ssdhip::decode_score_: expected (N, P, 4 + C) predictions");
    TORCH_CHECK_VALUE(dboxes.dim() == 2 && dboxes.size(0) == outputs.size(1) && dboxes.size(1) == 4, "ssdhip::decode_score_: default boxes must be (P, 4)");
    Ctx c(outputs);
    check(ssdhip_decode_score(outputs.data_ptr<float>(), dboxes.data_ptr<float>(), (int)outputs.size(0), (int)outputs.size(1), (int)(outputs.size(2) - 4), c.stream),
          "ssdhip::decode_score_");
}

// ---- A22 utils.non_maximum_suppression (utils.py:80-116), in place; returns the keep mask (N, P) uint8.  ws: caller-cached workspace
// of nms_ws_bytes(N, P) bytes (uint8) or an undefined tensor for the all-LDS variant (P <= 9000).
at::Tensor nms_(at::Tensor outputs, double iou_thresh, const c10::optional<at::Tensor>& ws, bool per_class) {
    need(outputs, "outputs");
    TORCH_CHECK_VALUE(outputs.dim() == 3 && outputs.size(2) >= 6, "ssdhip::nms_: expected (N, P, 4 + C) with C >= 2");
    const int N = (int)outputs.size(0), P = (int)outputs.size(1), C = (int)(outputs.size(2) - 4);
    at::Tensor keep = at::empty({N, P}, outputs.options().dtype(at::kByte));
    Ctx c(outputs);
    if (ws.has_value() && ws->defined()) {
        need(*ws, "ws", at::kByte);
        TORCH_CHECK_VALUE(ws->numel() >= ssdhip_nms_ws_bytes(N, P), "ssdhip::nms_: workspace smaller than nms_ws_bytes(N, P)");
        check((per_class ? ssdhip_nms_ws_per_class : ssdhip_nms_ws)(outputs.data_ptr<float>(), N, P, C, (float)iou_thresh, keep.data_ptr<uint8_t>(),
                                                                      ws->data_ptr(), c.stream), "ssdhip::nms_");
    } else {
        TORCH_CHECK_VALUE(!per_class, "ssdhip::nms_: per-class NMS runs on the workspace variant");
        check(ssdhip_nms(outputs.data_ptr<float>(), N, P, C, (float)iou_thresh, keep.data_ptr<uint8_t>(), c.stream), "ssdhip::nms_");
    }
    return keep;
}

int64_t nms_ws_bytes(int64_t N, int64_t P) { return ssdhip_nms_ws_bytes((int)N, (int)P); }

// ---- A21 utils.calc_iou (utils.py:58-77)
at::Tensor calc_iou(const at::Tensor& t, const at::Tensor& s) {
    need(t, "t"); need(s, "s");
    TORCH_CHECK_VALUE(t.dim() == 3 && s.dim() == 3 && t.size(0) == s.size(0) && t.size(2) >= 4 && s.size(2) >= 4, "ssdhip::calc_iou: bad shapes");
    at::Tensor out = at::empty({t.size(0), t.size(1), s.size(1)}, t.options());
    if (t.size(1) == 0 || s.size(1) == 0) return out;
    Ctx c(t);
    check(ssdhip_calc_iou(t.data_ptr<float>(), s.data_ptr<float>(), (int)t.size(0), (int)t.size(1), (int)s.size(1), (int)t.size(2), (int)s.size(2),
                          out.data_ptr<float>(), c.stream), "ssdhip::calc_iou");
    return out;
}

// ---- A17 torch.optim.Adam(lr, weight_decay) (train.py:97,122) as one launch over a device pointer table (optim.FusedAdam builds it)
void adam_multi_tensor_(const at::Tensor& table, const at::Tensor& sizes, const at::Tensor& blockmap, int64_t n_blocks, int64_t chunk,
                        at::Tensor step, at::Tensor hyper) {
    need(table, "table", at::kLong); need(sizes, "sizes", at::kLong); need(blockmap, "blockmap", at::kInt); need(step, "step", at::kInt); need(hyper, "hyper");
    Ctx c(table);
    check(ssdhip_adam_tick(step.data_ptr<int32_t>(), hyper.data_ptr<float>(), c.stream), "ssdhip::adam_multi_tensor_ (tick)");
    check(ssdhip_adam_multi_tensor((const void* const*)table.data_ptr<int64_t>(), sizes.data_ptr<int64_t>(), blockmap.data_ptr<int32_t>(), (int)n_blocks,
                                   (int)chunk, hyper.data_ptr<float>(), c.stream), "ssdhip::adam_multi_tensor_");
}

int64_t version() { return ssdhip_version(); }

}  // namespace

TORCH_LIBRARY(ssdhip, m) {
    m.def("version() -> int", &version);
    m.def("default_boxes(int[] maps, float s_min, float s_max, Device device) -> Tensor", &default_boxes);
    m.def("normalize_nchw_to_nhwc4(Tensor x) -> Tensor", &normalize_nchw_to_nhwc4);
    m.def("pack_conv_weight(Tensor w, int rows_pad, int inner_pad, int KP, int mode) -> Tensor", &pack_conv_weight);
    m.def("conv2d_nhwc(Tensor x, Tensor w_packed, Tensor? bias, Tensor? in_scale, Tensor? in_shift, int Cout, int k, int stride, int pad, int KP, int mode, "
          "int[] out_hw, bool want_stats, bool bf16) -> (Tensor, Tensor)", &conv2d_nhwc);
    m.def("conv2d_wgrad(Tensor x, Tensor dy, int[] w_shape, int stride, int pad, Tensor? in_scale, Tensor? in_shift, bool want_bias, int winograd, bool bf16) "
          "-> (Tensor, Tensor)", &conv2d_wgrad);
    m.def("bn_finalize(Tensor stats, int count, Tensor gamma, Tensor beta, Tensor(a!)? running_mean, Tensor(b!)? running_var, float momentum, float eps) "
          "-> (Tensor, Tensor, Tensor, Tensor)", &bn_finalize);
    m.def("bn_relu_pool(Tensor z, Tensor scale, Tensor shift, bool pool, int pool_pad) -> Tensor", &bn_relu_pool);
    m.def("multibox_loss_fwd(Tensor outputs, Tensor targets, Tensor default_bboxes, float alpha, float threshold) -> (Tensor, Tensor, Tensor)", &multibox_loss_fwd);
    m.def("multibox_loss_bwd(Tensor outputs, Tensor targets, Tensor default_bboxes, float alpha, float threshold, Tensor terms, Tensor sel, Tensor grad_loss) "
          "-> Tensor", &multibox_loss_bwd);
    m.def("decode_score_(Tensor(a!) outputs, Tensor default_bboxes) -> ()", &decode_score_);
    m.def("nms_(Tensor(a!) outputs, float iou_thresh, Tensor? ws, bool per_class) -> Tensor", &nms_);
    m.def("nms_ws_bytes(int N, int P) -> int", &nms_ws_bytes);
    m.def("calc_iou(Tensor t, Tensor s) -> Tensor", &calc_iou);
    m.def("adam_multi_tensor_(Tensor table, Tensor sizes, Tensor blockmap, int n_blocks, int chunk, Tensor(a!) step, Tensor(b!) hyper) -> ()", &adam_multi_tensor_);
}
