// Bilinear 2-D grid sampling for gfx950 (mode bilinear, zero padding, align_corners = False) and its adjoint.
//
// Replaces what torch_utils/ops/grid_sample_gradfix.py calls upstream: torch.nn.functional.grid_sample in the forward
// (grid_sample_gradfix.py:45) and aten::grid_sampler_2d_backward in the backward (:62-77).  G-NeRF never executes the op (its
// only user, the ADA augment pipe, is never constructed -- SURVEY F4; the renderer's own lookups live in the fused render
// kernel); it completes the custom-op surface.
//
// One lane per output pixel, lanes along the output row: the grid read (8 bytes per lane) and the output writes are
// coalesced, tap addresses and weights are computed once per pixel and reused for every channel; a second grid dimension
// splits the channels so that small images still fill the chip.  Forward is a gather (4 taps per channel and pixel);
// the adjoint scatters the image gradient with hardware float atomics (fp32 accumulation buffer, also for fp16 images)
// and sums the grid gradient per pixel in registers (channel chunks combine with one atomic per component).

#include "common.h"

namespace {

using namespace gnerf;

struct GsArgs {
    const void* image; const float* grid; void* out;               // forward: image [n,c,h,w], grid [n,ho,wo,2] fp32, out [n,c,ho,wo]
    const void* grad_out; float* grad_image; float* grad_grid;      // adjoint: grad_out like out; grad_image fp32 [n,c,h,w] (accumulated); grad_grid fp32 (accumulated)
    int n, c, h, w, ho, wo;
    int64_t is_n, is_c, is_h, is_w;                                 // image strides (elements)
    int c_chunk;
};

struct Taps { int o00, o01, o10, o11; int cx0, cx1, cy0, cy1; float w00, w01, w10, w11; float fx, fy; bool x0, x1, y0, y1; };

__device__ __forceinline__ Taps make_taps(float gx, float gy, const GsArgs& a) {
    float ix = ((gx + 1.f) * float(a.w) - 1.f) * 0.5f;
    float iy = ((gy + 1.f) * float(a.h) - 1.f) * 0.5f;
    // far-away (or non-finite) coordinates only ever produce zero taps: park them just outside the image
    ix = (ix > -2.f && ix < float(a.w) + 1.f) ? ix : -2.f;
    iy = (iy > -2.f && iy < float(a.h) + 1.f) ? iy : -2.f;
    const float x0f = floorf(ix), y0f = floorf(iy);
    Taps t;
    t.fx = ix - x0f; t.fy = iy - y0f;
    const int x0 = int(x0f), y0 = int(y0f), x1 = x0 + 1, y1 = y0 + 1;
    t.x0 = x0 >= 0 && x0 < a.w; t.x1 = x1 >= 0 && x1 < a.w; t.y0 = y0 >= 0 && y0 < a.h; t.y1 = y1 >= 0 && y1 < a.h;
    const int cx0 = min(max(x0, 0), a.w - 1), cx1 = min(max(x1, 0), a.w - 1), cy0 = min(max(y0, 0), a.h - 1), cy1 = min(max(y1, 0), a.h - 1);
    t.cx0 = cx0; t.cx1 = cx1; t.cy0 = cy0; t.cy1 = cy1;
    t.o00 = int(cy0 * a.is_h + cx0 * a.is_w); t.o01 = int(cy0 * a.is_h + cx1 * a.is_w);
    t.o10 = int(cy1 * a.is_h + cx0 * a.is_w); t.o11 = int(cy1 * a.is_h + cx1 * a.is_w);
    t.w00 = (t.x0 && t.y0) ? (1.f - t.fx) * (1.f - t.fy) : 0.f;
    t.w01 = (t.x1 && t.y0) ? t.fx * (1.f - t.fy) : 0.f;
    t.w10 = (t.x0 && t.y1) ? (1.f - t.fx) * t.fy : 0.f;
    t.w11 = (t.x1 && t.y1) ? t.fx * t.fy : 0.f;
    return t;
}

template <class T>
__global__ __launch_bounds__(256) void grid_sample_fwd_kernel(GsArgs a) {
    const int64_t pix = int64_t(blockIdx.x) * 256 + threadIdx.x;
    const int64_t npix = int64_t(a.n) * a.ho * a.wo;
    if (pix >= npix) return;
    const int img = int(pix / (int64_t(a.ho) * a.wo));
    const int64_t sp = pix - int64_t(img) * a.ho * a.wo;
    const float2 g = reinterpret_cast<const float2*>(a.grid)[pix];
    const Taps t = make_taps(g.x, g.y, a);
    const int c0 = blockIdx.y * a.c_chunk, c1 = min(a.c, c0 + a.c_chunk);
    const T* src = static_cast<const T*>(a.image) + img * a.is_n;
    T* dst = static_cast<T*>(a.out) + (int64_t(img) * a.c) * a.ho * a.wo + sp;
    for (int ch = c0; ch < c1; ch++) {
        const T* p = src + ch * a.is_c;
        const float v = load_as<T>(p, t.o00) * t.w00 + load_as<T>(p, t.o01) * t.w01 + load_as<T>(p, t.o10) * t.w10 + load_as<T>(p, t.o11) * t.w11;
        store_as<T>(dst, int64_t(ch) * a.ho * a.wo, v);
    }
}

template <class T>
__global__ __launch_bounds__(256) void grid_sample_bwd_kernel(GsArgs a) {
    const int64_t pix = int64_t(blockIdx.x) * 256 + threadIdx.x;
    const int64_t npix = int64_t(a.n) * a.ho * a.wo;
    if (pix >= npix) return;
    const int img = int(pix / (int64_t(a.ho) * a.wo));
    const int64_t sp = pix - int64_t(img) * a.ho * a.wo;
    const float2 g = reinterpret_cast<const float2*>(a.grid)[pix];
    const Taps t = make_taps(g.x, g.y, a);
    const int c0 = blockIdx.y * a.c_chunk, c1 = min(a.c, c0 + a.c_chunk);
    const T* src = a.image ? static_cast<const T*>(a.image) + img * a.is_n : nullptr;
    const T* go = static_cast<const T*>(a.grad_out) + (int64_t(img) * a.c) * a.ho * a.wo + sp;
    float* gi = a.grad_image ? a.grad_image + int64_t(img) * a.c * a.h * a.w : nullptr;
    float gix = 0.f, giy = 0.f;
    const bool v00 = t.x0 && t.y0, v01 = t.x1 && t.y0, v10 = t.x0 && t.y1, v11 = t.x1 && t.y1;
    // tap offsets inside the contiguous fp32 gradient image
    const int q00 = t.cy0 * a.w + t.cx0, q01 = t.cy0 * a.w + t.cx1, q10 = t.cy1 * a.w + t.cx0, q11 = t.cy1 * a.w + t.cx1;
    for (int ch = c0; ch < c1; ch++) {
        const float gv = load_as<T>(go, int64_t(ch) * a.ho * a.wo);
        if (gi) {
            float* pc = gi + int64_t(ch) * a.h * a.w;
            if (t.w00 != 0.f) unsafeAtomicAdd(pc + q00, gv * t.w00);
            if (t.w01 != 0.f) unsafeAtomicAdd(pc + q01, gv * t.w01);
            if (t.w10 != 0.f) unsafeAtomicAdd(pc + q10, gv * t.w10);
            if (t.w11 != 0.f) unsafeAtomicAdd(pc + q11, gv * t.w11);
        }
        if (a.grad_grid && src) {
            const T* p = src + ch * a.is_c;
            const float i00 = v00 ? load_as<T>(p, t.o00) : 0.f, i01 = v01 ? load_as<T>(p, t.o01) : 0.f;
            const float i10 = v10 ? load_as<T>(p, t.o10) : 0.f, i11 = v11 ? load_as<T>(p, t.o11) : 0.f;
            gix += gv * ((i01 - i00) * (1.f - t.fy) + (i11 - i10) * t.fy);
            giy += gv * ((i10 - i00) * (1.f - t.fx) + (i11 - i01) * t.fx);
        }
    }
    if (a.grad_grid) {
        float* gg = a.grad_grid + pix * 2;
        const float sx = gix * (0.5f * float(a.w)), sy = giy * (0.5f * float(a.h));
        if (gridDim.y == 1) { gg[0] += sx; gg[1] += sy; }
        else { unsafeAtomicAdd(gg, sx); unsafeAtomicAdd(gg + 1, sy); }
    }
}

int fill_common(GsArgs& a, int n, int c, int h, int w, const int64_t is[4], int ho, int wo, const char* what) {
    if (n < 1 || c < 1 || h < 1 || w < 1 || ho < 1 || wo < 1) return fail(GNERF_E_ARG, "%s: empty tensor", what);
    if (!is) return fail(GNERF_E_ARG, "%s: null strides", what);
    // tap offsets inside one image plane are 32-bit
    if ((int64_t(h) - 1) * is[2] + (int64_t(w) - 1) * is[3] >= (int64_t(1) << 31) || is[2] < 1 || is[3] < 1)
        return fail(GNERF_E_UNSUPPORTED, "%s: image plane too large or not positively strided", what);
    a.n = n; a.c = c; a.h = h; a.w = w; a.ho = ho; a.wo = wo;
    a.is_n = is[0]; a.is_c = is[1]; a.is_h = is[2]; a.is_w = is[3];
    const int64_t pix_blocks = (int64_t(n) * ho * wo + 255) / 256;
    int chunks = 1;
    while (pix_blocks * chunks < 4 * kNumCU && chunks < c) chunks *= 2;          // small images: split the channels over more workgroups
    a.c_chunk = (c + chunks - 1) / chunks;
    return GNERF_OK;
}

}  // namespace

extern "C" int gnerf_grid_sample_2d(const void* image, const float* grid, void* out, int dtype,
                                    int n, int c, int h, int w, const int64_t image_strides[4], int ho, int wo, gnerf_stream_t stream) {
    using namespace gnerf;
    if (!image || !grid || !out) return fail(GNERF_E_ARG, "grid_sample_2d: null pointer");
    if (dtype != GNERF_F32 && dtype != GNERF_F16) return fail(GNERF_E_UNSUPPORTED, "grid_sample_2d: float16/float32 only");
    GsArgs a{};
    if (int e = fill_common(a, n, c, h, w, image_strides, ho, wo, "grid_sample_2d")) return e;
    a.image = image; a.grid = grid; a.out = out;
    const dim3 grid_dim((unsigned)((int64_t(n) * ho * wo + 255) / 256), (unsigned)((c + a.c_chunk - 1) / a.c_chunk));
    if (dtype == GNERF_F32) hipLaunchKernelGGL(grid_sample_fwd_kernel<float>, grid_dim, dim3(256), 0, as_stream(stream), a);
    else hipLaunchKernelGGL(grid_sample_fwd_kernel<__half>, grid_dim, dim3(256), 0, as_stream(stream), a);
    return check_launch("grid_sample_2d");
}

extern "C" int gnerf_grid_sample_2d_backward(const void* grad_out, const void* image, const float* grid, float* grad_image, float* grad_grid,
                                             int dtype, int n, int c, int h, int w, const int64_t image_strides[4], int ho, int wo,
                                             gnerf_stream_t stream) {
    using namespace gnerf;
    if (!grad_out || !grid) return fail(GNERF_E_ARG, "grid_sample_2d_backward: null pointer");
    if (!grad_image && !grad_grid) return GNERF_OK;
    if (grad_grid && !image) return fail(GNERF_E_ARG, "grid_sample_2d_backward: the grid gradient needs the image");
    if (dtype != GNERF_F32 && dtype != GNERF_F16) return fail(GNERF_E_UNSUPPORTED, "grid_sample_2d_backward: float16/float32 only");
    GsArgs a{};
    if (int e = fill_common(a, n, c, h, w, image_strides, ho, wo, "grid_sample_2d_backward")) return e;
    a.image = image; a.grid = grid; a.grad_out = grad_out; a.grad_image = grad_image; a.grad_grid = grad_grid;
    const dim3 grid_dim((unsigned)((int64_t(n) * ho * wo + 255) / 256), (unsigned)((c + a.c_chunk - 1) / a.c_chunk));
    if (dtype == GNERF_F32) hipLaunchKernelGGL(grid_sample_bwd_kernel<float>, grid_dim, dim3(256), 0, as_stream(stream), a);
    else hipLaunchKernelGGL(grid_sample_bwd_kernel<__half>, grid_dim, dim3(256), 0, as_stream(stream), a);
    return check_launch("grid_sample_2d_backward");
}
