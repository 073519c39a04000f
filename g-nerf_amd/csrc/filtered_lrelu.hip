// filtered_lrelu_act_ for gfx950: in-place gain * leaky-ReLU * clamp with the bit-packed sign tensor.
//
// Replaces filtered_lrelu_plugin.filtered_lrelu_act_ (reference torch_utils/ops/filtered_lrelu.cpp:217,
// filtered_lrelu.cu:1109-1215), the helper behind the generic filtered_lrelu path
// (torch_utils/ops/filtered_lrelu.py:225-231).  G-NeRF never executes filtered_lrelu (only StyleGAN3's
// SynthesisLayer calls it and that class is never instantiated), so this op exists for API completeness.
//
// Sign tensor: uint8 [n, c, s_h, s_w/4], two bits per pixel (1 = value was negative, 2 = value was
// clamped), pixel x of a row in byte x>>2 at bit 2*(x&3).  Each lane owns four horizontally adjacent
// pixels, i.e. exactly one sign byte: no cross-lane packing is needed on a 64-wide wave.

#include "common.h"

namespace {

using namespace gnerf;

struct ActArgs {
    void* x; uint8_t* s;
    int n, c, h, w;
    int64_t xs_n, xs_c, xs_h, xs_w;
    int s_h, s_w, sx, sy;
    float gain, slope, clamp;
};

// MODE 0: no signs; 1: write signs; 2: read signs
template <class T, int MODE>
__global__ __launch_bounds__(256) void lrelu_act_kernel(ActArgs a) {
    typedef typename Arith<T>::type A;
    T* x = static_cast<T*>(a.x);
    const int row_w = (MODE == 1) ? a.s_w : a.w;             // logical row width walked by the grid
    const int quads = (row_w + 3) >> 2;
    const int rows = (MODE == 1) ? a.s_h : a.h;
    const int64_t total = int64_t(a.n) * a.c * rows * quads;
    const int64_t stride = int64_t(gridDim.x) * 256;
    const A gain = A(a.gain), slope = A(a.slope), clamp = A(a.clamp);
    for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < total; i += stride) {
        const int q = int(i % quads);
        int64_t r = i / quads;
        const int y = int(r % rows); r /= rows;
        const int ch = int(r % a.c);
        const int img = int(r / a.c);
        T* px = x + img * a.xs_n + ch * a.xs_c + y * a.xs_h;
        unsigned bits = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int xx = 4 * q + k;
            if (xx >= a.w || y >= a.h) continue;
            A v = load_as<T>(px, xx * a.xs_w) * gain;
            if (MODE == 2) {
                const unsigned sxx = unsigned(xx + a.sx), syy = unsigned(y + a.sy);
                if (sxx < unsigned(a.s_w) && syy < unsigned(a.s_h)) {
                    const uint64_t is = (sxx >> 2) + uint64_t(a.s_w >> 2) * (syy + uint64_t(a.s_h) * (uint64_t(img) * a.c + ch));
                    const unsigned sb = unsigned(a.s[is]) >> ((sxx & 3) << 1);
                    if (sb & 1) v *= slope;
                    if (sb & 2) v = A(0);
                }
            } else {
                unsigned sgn = 0;
                if (v < A(0)) { v *= slope; sgn = 1; }
                if (fabs(double(v)) > double(clamp)) { v = v < A(0) ? -clamp : clamp; sgn = 2; }
                bits |= sgn << (2 * k);
            }
            store_as<T>(px, xx * a.xs_w, v);
        }
        if (MODE == 1) {
            const uint64_t is = uint64_t(q) + uint64_t(a.s_w >> 2) * (y + uint64_t(a.s_h) * (uint64_t(img) * a.c + ch));
            a.s[is] = uint8_t(bits);
        }
    }
}

template <class T>
int launch_act_mode(const ActArgs& a, int mode, hipStream_t stream) {
    const int row_w = (mode == 1) ? a.s_w : a.w, rows = (mode == 1) ? a.s_h : a.h;
    const int64_t total = int64_t(a.n) * a.c * rows * ((row_w + 3) >> 2);
    int64_t blocks = (total + 255) / 256;
    if (blocks > int64_t(kNumCU) * 8) blocks = int64_t(kNumCU) * 8;
    if (blocks < 1) blocks = 1;
    dim3 g((unsigned)blocks), t(256);
    if (mode == 0) hipLaunchKernelGGL((lrelu_act_kernel<T, 0>), g, t, 0, stream, a);
    else if (mode == 1) hipLaunchKernelGGL((lrelu_act_kernel<T, 1>), g, t, 0, stream, a);
    else hipLaunchKernelGGL((lrelu_act_kernel<T, 2>), g, t, 0, stream, a);
    return check_launch("filtered_lrelu_act");
}

}  // namespace

extern "C" int gnerf_filtered_lrelu_act(void* x, uint8_t* s, int dtype, int n, int c, int h, int w,
                                        const int64_t xs[4], int s_h, int s_w, int sx, int sy,
                                        float gain, float slope, float clamp, int mode, gnerf_stream_t stream) {
    using namespace gnerf;
    if (!x || !xs) return fail(GNERF_E_ARG, "filtered_lrelu_act: null pointer");
    if (n < 1 || c < 1 || h < 1 || w < 1) return fail(GNERF_E_ARG, "filtered_lrelu_act: x is empty");
    if (mode < 0 || mode > 2) return fail(GNERF_E_ARG, "filtered_lrelu_act: mode must be 0, 1 or 2");
    if (mode != 0) {
        if (!s) return fail(GNERF_E_ARG, "filtered_lrelu_act: sign tensor is null");
        if (s_h < 1 || s_w < 4 || (s_w & 3)) return fail(GNERF_E_ARG, "filtered_lrelu_act: bad sign tensor shape %dx%d", s_h, s_w);
        if (mode == 1 && (s_h < h || s_w < w)) return fail(GNERF_E_ARG, "filtered_lrelu_act: sign tensor smaller than x");
    }
    ActArgs a{x, s, n, c, h, w, xs[0], xs[1], xs[2], xs[3], s_h, s_w, sx, sy, gain, slope, clamp};
    hipStream_t st = as_stream(stream);
    switch (dtype) {
        case GNERF_F32: return launch_act_mode<float>(a, mode, st);
        case GNERF_F16: return launch_act_mode<__half>(a, mode, st);
        case GNERF_F64: return launch_act_mode<double>(a, mode, st);
        default: return fail(GNERF_E_ARG, "filtered_lrelu_act: unsupported dtype code %d", dtype);
    }
}
