// gnerf_torch_ext: the thin PyTorch-ROCm C++ extension over libgnerf_hip.so's C ABI (include/gnerf_hip.h).
//
// The reference loads three pybind plugins (torch_utils/custom_ops.py:61-157) whose entry points are
//   bias_act_plugin.bias_act                      torch_utils/ops/bias_act.cpp:36, :98-101
//   upfirdn2d_plugin.upfirdn2d                    torch_utils/ops/upfirdn2d.cpp:20, :106-109
//   filtered_lrelu_plugin.filtered_lrelu          torch_utils/ops/filtered_lrelu.cpp:20
//   filtered_lrelu_plugin.filtered_lrelu_act_     torch_utils/ops/filtered_lrelu.cpp:217, :298-302
// This module exports functions with exactly those names, argument lists and conventions (absent tensors are 0-element
// tensors, outputs are allocated here with torch::empty, the launch goes to the current stream of x's device under a device
// guard) -- and nothing else: no kernels live here, every function validates, allocates and calls the C ABI.  It exists for host
// cost: a call through ctypes spends 10-12 us marshalling arguments in Python, this path ~3.
// Also exported: render_forward (the fused renderer has no reference plugin; same C ABI call as gnerf_hip.render_forward).
//
// Built ahead of time by csrc/build.sh (g++, no hipcc: there is no device code) into g-nerf_amd/gnerf_hip/gnerf_torch_ext.so.

#include <torch/extension.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>

#include <tuple>
#include <vector>

#include "gnerf_hip.h"

namespace {

using torch::Tensor;

int dtype_code(const Tensor& t, const char* what) {
    switch (t.scalar_type()) {
        case torch::kFloat32: return GNERF_F32;
        case torch::kFloat16: return GNERF_F16;
        case torch::kFloat64: return GNERF_F64;
        default: TORCH_CHECK(false, what, ": unsupported dtype ", t.scalar_type());
    }
    return -1;
}

inline bool present(const Tensor& t) { return t.defined() && t.numel() > 0; }
inline void* ptr_or_null(const Tensor& t) { return present(t) ? t.data_ptr() : nullptr; }

inline gnerf_stream_t current_stream() { return c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream(); }

inline void check_rc(int rc, const char* what) { TORCH_CHECK(rc == GNERF_OK, what, " failed (", rc, "): ", gnerf_last_error()); }

// strides compared only where the size is >= 2 (bias_act.cpp:18-29)
bool same_layout(const Tensor& a, const Tensor& b) {
    if (a.dim() != b.dim()) return false;
    for (int64_t i = 0; i < a.dim(); i++) {
        if (a.size(i) != b.size(i)) return false;
        if (a.size(i) >= 2 && a.stride(i) != b.stride(i)) return false;
    }
    return true;
}

void strides4(const Tensor& t, int64_t (&s)[4]) {
    for (int i = 0; i < 4; i++) s[i] = t.stride(i);
}

// ------------------------------------------------------------------------------------------------ bias_act.cpp:36

Tensor bias_act(Tensor x, Tensor b, Tensor xref, Tensor yref, Tensor dy, int grad, int dim, int act, float alpha, float gain, float clamp) {
    TORCH_CHECK(x.is_cuda(), "x must reside on CUDA device");
    TORCH_CHECK(!present(b) || (b.dtype() == x.dtype() && b.device() == x.device()), "b must have the same dtype and device as x");
    TORCH_CHECK(!present(xref) || (xref.sizes() == x.sizes() && xref.dtype() == x.dtype() && xref.device() == x.device()), "xref must have the same shape, dtype, and device as x");
    TORCH_CHECK(!present(yref) || (yref.sizes() == x.sizes() && yref.dtype() == x.dtype() && yref.device() == x.device()), "yref must have the same shape, dtype, and device as x");
    TORCH_CHECK(!present(dy) || (dy.sizes() == x.sizes() && dy.dtype() == x.dtype() && dy.device() == x.device()), "dy must have the same dtype and device as x");
    TORCH_CHECK(x.numel() <= INT_MAX, "x is too large");
    TORCH_CHECK(!present(b) || b.dim() == 1, "b must have rank 1");
    TORCH_CHECK(!present(b) || (dim >= 0 && dim < x.dim()), "dim is out of bounds");
    TORCH_CHECK(!present(b) || b.numel() == x.size(dim), "b has wrong number of elements");
    TORCH_CHECK(grad >= 0, "grad must be non-negative");
    TORCH_CHECK(x.is_non_overlapping_and_dense(), "x must be non-overlapping and dense");
    TORCH_CHECK(!present(b) || b.is_contiguous(), "b must be contiguous");
    TORCH_CHECK(!present(xref) || same_layout(xref, x), "xref must have the same layout as x");
    TORCH_CHECK(!present(yref) || same_layout(yref, x), "yref must have the same layout as x");
    TORCH_CHECK(!present(dy) || same_layout(dy, x), "dy must have the same layout as x");
    const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(at::device_of(x));
    Tensor y = torch::empty_like(x);
    TORCH_CHECK(same_layout(y, x), "y must have the same layout as x");
    const int size_b = present(b) ? int(b.numel()) : 0;
    const int64_t step_b = present(b) ? x.stride(dim) : 1;
    check_rc(gnerf_bias_act(x.data_ptr(), ptr_or_null(b), ptr_or_null(xref), ptr_or_null(yref), ptr_or_null(dy), y.data_ptr(),
                            dtype_code(x, "bias_act"), x.numel(), size_b, step_b, grad, act, alpha, gain, clamp, current_stream()),
             "gnerf_bias_act");
    return y;
}

// ------------------------------------------------------------------------------------------------ upfirdn2d.cpp:20

Tensor upfirdn2d(Tensor x, Tensor f, int upx, int upy, int downx, int downy, int padx0, int padx1, int pady0, int pady1, bool flip, float gain) {
    TORCH_CHECK(x.is_cuda(), "x must reside on CUDA device");
    TORCH_CHECK(f.device() == x.device(), "f must reside on the same device as x");
    TORCH_CHECK(f.dtype() == torch::kFloat, "f must be float32");
    TORCH_CHECK(x.numel() <= INT_MAX, "x is too large");
    TORCH_CHECK(f.numel() <= INT_MAX, "f is too large");
    TORCH_CHECK(x.numel() > 0, "x has zero size");
    TORCH_CHECK(f.numel() > 0, "f has zero size");
    TORCH_CHECK(x.dim() == 4, "x must be rank 4");
    TORCH_CHECK(f.dim() == 2, "f must be rank 2");
    TORCH_CHECK(f.size(0) >= 1 && f.size(1) >= 1, "f must be at least 1x1");
    TORCH_CHECK(upx >= 1 && upy >= 1, "upsampling factor must be at least 1");
    TORCH_CHECK(downx >= 1 && downy >= 1, "downsampling factor must be at least 1");
    const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(at::device_of(x));
    const int64_t out_w = (x.size(3) * upx + padx0 + padx1 - f.size(1) + downx) / downx;
    const int64_t out_h = (x.size(2) * upy + pady0 + pady1 - f.size(0) + downy) / downy;
    TORCH_CHECK(out_w >= 1 && out_h >= 1, "output must be at least 1x1");
    Tensor y = torch::empty({x.size(0), x.size(1), out_h, out_w}, x.options(), x.suggest_memory_format());
    TORCH_CHECK(y.numel() <= INT_MAX, "output is too large");
    int64_t xs[4], ys[4], fs[2] = {f.stride(0), f.stride(1)};
    strides4(x, xs);
    strides4(y, ys);
    check_rc(gnerf_upfirdn2d(x.data_ptr(), f.data_ptr<float>(), y.data_ptr(), dtype_code(x, "upfirdn2d"),
                             int(x.size(0)), int(x.size(1)), int(x.size(2)), int(x.size(3)), xs, int(f.size(0)), int(f.size(1)), fs,
                             int(out_h), int(out_w), ys, upx, upy, downx, downy, padx0, pady0, flip ? 1 : 0, gain, current_stream()),
             "gnerf_upfirdn2d");
    return y;
}

// ------------------------------------------------------------------------------------------------ filtered_lrelu.cpp:20

std::tuple<Tensor, Tensor, int> filtered_lrelu(Tensor x, Tensor fu, Tensor fd, Tensor b, Tensor si, int up, int down, int px0, int px1, int py0, int py1,
                                                int sx, int sy, float gain, float slope, float clamp, bool flip_filters, bool writeSigns) {
    TORCH_CHECK(x.is_cuda(), "x must reside on CUDA device");
    TORCH_CHECK(fu.device() == x.device() && fd.device() == x.device() && b.device() == x.device(), "all input tensors must reside on the same device");
    TORCH_CHECK(fu.dtype() == torch::kFloat && fd.dtype() == torch::kFloat, "fu and fd must be float32");
    TORCH_CHECK(b.dtype() == x.dtype(), "x and b must have the same dtype");
    TORCH_CHECK(x.dim() == 4 && x.numel() > 0, "x must be a non-empty rank-4 tensor");
    TORCH_CHECK((fu.dim() == 1 || fu.dim() == 2) && (fd.dim() == 1 || fd.dim() == 2), "fu and fd must be rank 1 or 2");
    TORCH_CHECK(fu.numel() > 0 && fd.numel() > 0, "fu and fd must be non-empty");
    TORCH_CHECK(b.dim() == 1 && b.size(0) == x.size(1), "b must be a vector with the same number of channels as x");
    TORCH_CHECK(up >= 1 && down >= 1, "up and down must be at least 1");
    const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(at::device_of(x));
    const auto none = [&]() { return std::make_tuple(Tensor(torch::empty({0}, x.options())), Tensor(torch::empty({0}, x.options())), -1); };
    if (x.scalar_type() != torch::kFloat32 && x.scalar_type() != torch::kFloat16) return none();       // no fused kernel: caller's generic route
    if ((fu.dim() == 2 && !(fu.size(0) == 1 && fu.size(1) == 1)) || (fd.dim() == 2 && !(fd.size(0) == 1 && fd.size(1) == 1))) return none();
    const int64_t fut = fu.size(-1) - 1, fdt = fd.size(-1) - 1;
    const int64_t cw = x.size(3) * up + (px0 + px1) - fut, ch = x.size(2) * up + (py0 + py1) - fut;
    TORCH_CHECK(cw > fdt && ch > fdt, "upsampled buffer must be at least the size of downsampling filter");
    const int64_t yw = (cw - fdt + (down - 1)) / down, yh = (ch - fdt + (down - 1)) / down;
    TORCH_CHECK(yw >= 1 && yh >= 1, "output must be at least 1x1");
    Tensor y = torch::empty({x.size(0), x.size(1), yh, yw}, x.options(), x.suggest_memory_format());
    const bool readSigns = present(si);
    Tensor so = torch::empty({0}, x.options().dtype(torch::kUInt8));
    Tensor s;
    int64_t s_h = 0, s_w = 0;
    int mode = 0;
    if (writeSigns) {
        s_h = yh * down - (down - 1) + fdt;
        s_w = (yw * down - (down - 1) + fdt + 15) & ~int64_t(15);
        so = torch::empty({x.size(0), x.size(1), s_h, s_w >> 2}, x.options().dtype(torch::kUInt8));
        s = so;
        mode = 1;
    } else if (readSigns) {
        TORCH_CHECK(si.is_cuda() && si.scalar_type() == torch::kUInt8 && si.dim() == 4 && si.is_contiguous() && si.size(0) == x.size(0) && si.size(1) == x.size(1),
                    "signs must be a contiguous uint8 [n, c, h, w/4] tensor matching x");
        s = si;
        s_h = si.size(2);
        s_w = si.size(3) * 4;
        mode = 2;
    }
    Tensor fu_c = fu.contiguous(), fd_c = fd.contiguous(), b_c = b.contiguous();
    int64_t xs[4], ys[4];
    strides4(x, xs);
    strides4(y, ys);
    const int rc = gnerf_filtered_lrelu(x.data_ptr(), fu_c.data_ptr<float>(), fd_c.data_ptr<float>(), b_c.data_ptr(),
                                        s.defined() ? s.data_ptr<uint8_t>() : nullptr, y.data_ptr(), dtype_code(x, "filtered_lrelu"),
                                        int(x.size(0)), int(x.size(1)), int(x.size(2)), int(x.size(3)), xs, int(yh), int(yw), ys,
                                        int(fu.size(-1)), int(fu.dim()), int(fd.size(-1)), int(fd.dim()), up, down, px0, py0,
                                        int(s_h), int(s_w), sx, sy, mode, gain, slope, clamp, flip_filters ? 1 : 0, current_stream());
    if (rc == GNERF_E_UNSUPPORTED) return none();
    check_rc(rc, "gnerf_filtered_lrelu");
    return std::make_tuple(y, so, 0);
}

// ------------------------------------------------------------------------------------------------ filtered_lrelu.cpp:217

Tensor filtered_lrelu_act_(Tensor x, Tensor si, int sx, int sy, float gain, float slope, float clamp, bool writeSigns) {
    TORCH_CHECK(x.is_cuda(), "x must reside on CUDA device");
    TORCH_CHECK(x.dim() == 4, "x must be rank 4");
    const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(at::device_of(x));
    const bool readSigns = present(si);
    Tensor so = torch::empty({0}, x.options().dtype(torch::kUInt8));
    Tensor s;
    int64_t s_h = 0, s_w = 0;
    int mode = 0;
    if (readSigns) {
        TORCH_CHECK(si.is_cuda() && si.scalar_type() == torch::kUInt8 && si.dim() == 4 && si.is_contiguous(), "si must be a contiguous rank-4 uint8 tensor");
        s = si;
        s_h = si.size(2);
        s_w = si.size(3) * 4;
        mode = 2;
    } else if (writeSigns) {
        s_w = (x.size(3) + 15) & ~int64_t(15);
        s_h = x.size(2);
        so = torch::empty({x.size(0), x.size(1), s_h, s_w >> 2}, x.options().dtype(torch::kUInt8));
        s = so;
        mode = 1;
        sx = sy = 0;
    }
    int64_t xs[4];
    strides4(x, xs);
    check_rc(gnerf_filtered_lrelu_act(x.data_ptr(), s.defined() ? s.data_ptr<uint8_t>() : nullptr, dtype_code(x, "filtered_lrelu_act_"),
                                      int(x.size(0)), int(x.size(1)), int(x.size(2)), int(x.size(3)), xs, int(s_h), int(s_w), sx, sy,
                                      gain, slope, clamp, mode, current_stream()),
             "gnerf_filtered_lrelu_act");
    return so;
}

// ------------------------------------------------------------------------------------------------ the fused renderer

// planes_nhwc [3N,H,W,32] (or interleaved [N,H,W,96]) f32 contiguous; w1,b1,w2,b2 effective f32 weights; rays [N,M,3]; noise_coarse [N*M*S]; noise_fine [N*M*F] or
// empty; ray_start_t / ray_end_t per-ray limits or empty (then the scalars are used); planes_absmax one float or empty; workspace uint8
// (zeroed once by the caller).  Returns (rgb [N,M,32], depth [N,M,1], wsum [N,M,1]).
std::tuple<Tensor, Tensor, Tensor> render_forward(Tensor planes_nhwc, int64_t n_items, Tensor w1, Tensor b1, Tensor w2, Tensor b2,
                                                  Tensor ray_origins, Tensor ray_dirs, Tensor noise_coarse, Tensor noise_fine,
                                                  int64_t depth_resolution, int64_t depth_resolution_importance, double ray_start, double ray_end,
                                                  Tensor ray_start_t, Tensor ray_end_t, double box_warp, bool white_back, bool disparity_space_sampling,
                                                  int64_t image_width, Tensor planes_absmax, int64_t mlp_mode, Tensor workspace,
                                                  bool planes_shared, bool depth_clamp_per_item) {
    auto f32c = [](const Tensor& t, const char* name) {
        TORCH_CHECK(t.is_cuda() && t.scalar_type() == torch::kFloat32 && t.is_contiguous(), "render_forward: ", name, " must be a contiguous float32 GPU tensor");
    };
    f32c(planes_nhwc, "planes_nhwc"); f32c(w1, "w1"); f32c(b1, "b1"); f32c(w2, "w2"); f32c(b2, "b2");
    f32c(ray_origins, "ray_origins"); f32c(ray_dirs, "ray_dirs"); f32c(noise_coarse, "noise_coarse");
    const int64_t plane_items = planes_shared ? 1 : n_items;   // planes_shared: one item's planes read by all n_items items of rays
    const bool separate = planes_nhwc.dim() == 4 && planes_nhwc.size(3) == 32 && planes_nhwc.size(0) == 3 * plane_items;
    const bool interleaved = planes_nhwc.dim() == 4 && planes_nhwc.size(3) == 96 && planes_nhwc.size(0) == plane_items;
    TORCH_CHECK(separate || interleaved, "render_forward: planes_nhwc must be [3N,H,W,32] or [N,H,W,96] (N = 1 when planes_shared)");
    TORCH_CHECK(w1.numel() == 64 * 32 && b1.numel() == 64 && w2.numel() == 33 * 64 && b2.numel() == 33, "render_forward: decoder must be the 32->64->33 MLP");
    TORCH_CHECK(ray_origins.dim() == 3 && ray_origins.size(0) == n_items && ray_origins.size(2) == 3 && ray_dirs.sizes() == ray_origins.sizes(), "render_forward: rays must be [N,M,3]");
    const int64_t m = ray_origins.size(1), S = depth_resolution, F = depth_resolution_importance;
    TORCH_CHECK(noise_coarse.numel() == n_items * m * S, "render_forward: noise_coarse must have N*M*S elements");
    if (F > 0) { f32c(noise_fine, "noise_fine"); TORCH_CHECK(noise_fine.numel() == n_items * m * F, "render_forward: noise_fine must have N*M*F elements"); }
    if (present(ray_start_t)) {
        f32c(ray_start_t, "ray_start"); f32c(ray_end_t, "ray_end");
        TORCH_CHECK(ray_start_t.numel() == n_items * m && ray_end_t.numel() == n_items * m, "render_forward: per-ray limits must have N*M elements");
    }
    if (present(planes_absmax)) { f32c(planes_absmax, "planes_absmax"); TORCH_CHECK(planes_absmax.numel() == 1, "render_forward: planes_absmax must have one element"); }
    TORCH_CHECK(workspace.is_cuda() && workspace.is_contiguous() && size_t(workspace.numel() * workspace.element_size()) >= gnerf_render_workspace_bytes(),
                "render_forward: workspace too small");
    const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(at::device_of(planes_nhwc));
    auto opts = planes_nhwc.options();
    Tensor rgb = torch::empty({n_items, m, 32}, opts), depth = torch::empty({n_items, m, 1}, opts), wsum = torch::empty({n_items, m, 1}, opts);
    gnerf_render_params p = {};
    p.planes_nhwc = planes_nhwc.data_ptr<float>(); p.n_items = int32_t(n_items); p.plane_h = int32_t(planes_nhwc.size(1)); p.plane_w = int32_t(planes_nhwc.size(2));
    p.ray_origins = ray_origins.data_ptr<float>(); p.ray_dirs = ray_dirs.data_ptr<float>(); p.rays_per_item = int32_t(m); p.image_width = int32_t(image_width);
    p.w1 = w1.data_ptr<float>(); p.b1 = b1.data_ptr<float>(); p.w2 = w2.data_ptr<float>(); p.b2 = b2.data_ptr<float>();
    p.depth_resolution = int32_t(S); p.depth_resolution_importance = int32_t(F);
    p.ray_start = float(ray_start); p.ray_end = float(ray_end);
    p.ray_start_per_ray = present(ray_start_t) ? ray_start_t.data_ptr<float>() : nullptr;
    p.ray_end_per_ray = present(ray_start_t) ? ray_end_t.data_ptr<float>() : nullptr;
    p.box_warp = float(box_warp); p.white_back = white_back ? 1 : 0; p.disparity_space_sampling = disparity_space_sampling ? 1 : 0;
    p.noise_coarse = noise_coarse.data_ptr<float>(); p.noise_fine = F > 0 ? noise_fine.data_ptr<float>() : nullptr;
    p.out_rgb = rgb.data_ptr<float>(); p.out_depth = depth.data_ptr<float>(); p.out_wsum = wsum.data_ptr<float>();
    p.workspace = workspace.data_ptr(); p.debug = nullptr;
    p.planes_absmax = present(planes_absmax) ? planes_absmax.data_ptr<float>() : nullptr;
    p.mlp_mode = int32_t(mlp_mode);
    p.planes_interleaved = interleaved ? 1 : 0;
    p.planes_shared = planes_shared ? 1 : 0; p.depth_clamp_per_item = depth_clamp_per_item ? 1 : 0;
    check_rc(gnerf_render_forward(&p, current_stream()), "gnerf_render_forward");
    return std::make_tuple(rgb, depth, wsum);
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.def("bias_act", &bias_act);
    m.def("upfirdn2d", &upfirdn2d);
    m.def("filtered_lrelu", &filtered_lrelu);
    m.def("filtered_lrelu_act_", &filtered_lrelu_act_);
    m.def("render_forward", &render_forward);
    // the header version THIS extension was compiled against (a compile-time constant: gnerf_abi_version() would resolve in
    // libgnerf_hip.so at run time and compare the library with itself); gnerf_hip.ext() checks both against its own
    m.def("abi_version", []() { return int(GNERF_ABI_VERSION); });
    m.def("library_abi_version", []() { return gnerf_abi_version(); });
}
