// The surroundings of the modulated convolutions: everything StyleGAN2's SynthesisLayer / ToRGBLayer do
// around the MIOpen convolution itself (networks_stylegan2.py:41-98 modulated_conv2d, :315-334 SynthesisLayer.forward), which
// upstream is a chain of ~10 ATen elementwise / reduction launches per layer:
//
//   gnerf_modulate_weights   per-sample weights w[n,o,i,k] = weight[o,i,k] * styles[n,i] * dcoef[n,o]  (fused form, :61-75) and /
//                            or the demodulation coefficients dcoef[n,o] = rsqrt(sum_ik (weight styles)^2 + 1e-8) alone (un-fused
//                            form, :71-72), with the fp16 pre-normalisation of :62-64 -- one launch instead of 7.
//   gnerf_scale_channels     x[n,c,:,:] * styles[n,c]  (the un-fused form's input scaling, :77).
//   gnerf_modconv_epilogue   what follows the convolution: demodulation (un-fused, :79-80) + noise (:79-83 / :96-97) + bias +
//                            activation + gain + clamp (bias_act, :331-333) in ONE pass over the activations instead of three.
// Every intermediate the composed ops would have materialised in the activations' dtype is rounded to that dtype here too, so the
// results equal the composed ops' up to a last-place rounding (tests/test_gpu_parity.py::test_modconv_*).

#include "common.h"

namespace {

using namespace gnerf;

constexpr int kThreads = 256;

__device__ __forceinline__ float block_reduce(float v, bool is_max, float* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float w = __shfl_xor(v, o);
        v = is_max ? fmaxf(v, w) : v + w;
    }
    __syncthreads();                                   // `red` may still be read from a previous reduction
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    const float a = red[0], b = red[1], c = red[2], d = red[3];
    return is_max ? fmaxf(fmaxf(a, b), fmaxf(c, d)) : (a + b) + (c + d);
}

// One workgroup per (n, o).  ik = I * K * K elements of weight[o] against styles[n, i = e / kk].
template <class T>
__global__ __launch_bounds__(kThreads) void modulate_weights_kernel(const float* __restrict__ weight, const float* __restrict__ styles,
                                                                    T* __restrict__ out, float* __restrict__ dcoefs,
                                                                    int n_out, int n_in, int kk, int demodulate, int prenorm, int layout) {
    __shared__ float red[4];
    const int o = blockIdx.x % n_out, n = blockIdx.x / n_out;
    const int ik = n_in * kk;
    const float* w = weight + int64_t(o) * ik;
    const float* s = styles + int64_t(n) * n_in;
    float w_scale = 1.f, s_scale = 1.f;
    if (prenorm) {                                     // networks_stylegan2.py:62-64: weight / (sqrt(fan_in) max|weight[o]|), styles / max|styles[n]|
        float wm = 0.f, sm = 0.f;
        for (int e = threadIdx.x; e < ik; e += kThreads) wm = fmaxf(wm, fabsf(w[e]));
        for (int i = threadIdx.x; i < n_in; i += kThreads) sm = fmaxf(sm, fabsf(s[i]));
        wm = block_reduce(wm, true, red);
        sm = block_reduce(sm, true, red);
        w_scale = (1.0f / sqrtf(float(ik))) / wm;
        s_scale = 1.0f / sm;
    }
    float d = 1.f;
    if (demodulate) {
        float acc = 0.f;
        for (int e = threadIdx.x; e < ik; e += kThreads) {
            const float v = (w[e] * w_scale) * (s[e / kk] * s_scale);
            acc = fmaf(v, v, acc);
        }
        d = rsqrtf(block_reduce(acc, false, red) + 1e-8f);
        if (dcoefs && threadIdx.x == 0) dcoefs[int64_t(n) * n_out + o] = d;
    }
    if (out) {
        // element (o, i, k) of sample n at o * so + i * si + k * sk: [O,I,K] (conv2d), [I,O,K] (conv_transpose2d), and the
        // channels_last memory of either, [O,K,I] / [I,K,O]
        const int64_t so = layout == 0 ? ik : (layout == 1 ? kk : (layout == 2 ? ik : 1));
        const int64_t si = layout == 0 ? kk : (layout == 1 ? int64_t(n_out) * kk : (layout == 2 ? 1 : int64_t(n_out) * kk));
        const int64_t sk = layout == 0 || layout == 1 ? 1 : (layout == 2 ? n_in : n_out);
        T* dst = out + int64_t(n) * n_out * ik + o * so;
        for (int e = threadIdx.x; e < ik; e += kThreads) {
            const int i = e / kk, k = e - i * kk;
            store_as<T>(dst, i * si + k * sk, ((w[e] * w_scale) * (s[i] * s_scale)) * d);
        }
    }
}

// fp16 pre-normalised styles alone (the un-fused form multiplies the activations by them): styles[n,:] / max|styles[n,:]|
__global__ __launch_bounds__(kThreads) void normalise_styles_kernel(const float* __restrict__ styles, float* __restrict__ out, int n_in) {
    __shared__ float red[4];
    const float* s = styles + int64_t(blockIdx.x) * n_in;
    float sm = 0.f;
    for (int i = threadIdx.x; i < n_in; i += kThreads) sm = fmaxf(sm, fabsf(s[i]));
    sm = block_reduce(sm, true, red);
    for (int i = threadIdx.x; i < n_in; i += kThreads) out[int64_t(blockIdx.x) * n_in + i] = s[i] / sm;
}


struct EpiArgs {
    const void* x; void* y;
    const float* scale;      // [rows] or NULL
    const float* noise;      // [row_len] (or [rows / channels][row_len] when noise_per_item) or NULL
    const void* bias;        // [channels], activations' dtype, or NULL
    unsigned row_len, channels;
    int noise_per_item, round_noise, act;
    float alpha, gain, clamp;
};

// y[row, :] = clamp(act(round_T(x[row, :] * T(scale[row]) + noise[:]) + bias[row % channels]) * gain); rows = blockIdx.y
// SCALE / NOISE are compile-time so that the superresolution's layers (no noise) and the fused-convolution form (no scale) carry
// none of the other's instructions: the fp16 kernel is instruction-bound (first version, everything at run time: 2.2 TB/s on
// [4,128,512,512] against bias_act's 5.5).  The fp16 demodulation + noise step is ONE packed half FMA per pair of elements:
// x, scale and noise are halves there, so v_pk_fma_f16 rounds the exact x * scale + noise once -- the value torch.addcmul / the
// half multiply of the composed ops store.
template <class T, int VEC, int ACT, bool SCALE, bool NOISE>
__global__ __launch_bounds__(kThreads) void modconv_epilogue_kernel(EpiArgs a) {
    typedef Pk<T, VEC> P;
    const unsigned row = blockIdx.y;
    const unsigned v = blockIdx.x * kThreads + threadIdx.x;
    if (v * VEC >= a.row_len) return;
    const T* x = static_cast<const T*>(a.x) + int64_t(row) * a.row_len;
    T* y = static_cast<T*>(a.y) + int64_t(row) * a.row_len;
    const float bv = a.bias ? float(load_as<T>(static_cast<const T*>(a.bias), row % a.channels)) : 0.f;
    const float* nz = NOISE ? a.noise + (a.noise_per_item ? int64_t(row / a.channels) * a.row_len : 0) : nullptr;
    P in = *reinterpret_cast<const P*>(x + v * VEC);
    float nv[VEC];
    if constexpr (NOISE) {
#pragma unroll
        for (int k = 0; k < VEC; k++) nv[k] = nz[v * VEC + k];
    }
    float t[VEC];
    if constexpr (sizeof(T) == 2 && VEC % 2 == 0 && (SCALE || NOISE)) {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        const _Float16 sh = SCALE ? (_Float16)a.scale[row] : (_Float16)1.0f;
        const h2 s2 = {sh, sh};
#pragma unroll
        for (int k = 0; k < VEC; k += 2) {
            h2 xv = {__builtin_bit_cast(_Float16, in.v[k]), __builtin_bit_cast(_Float16, in.v[k + 1])};
            h2 r;
            if constexpr (NOISE) {
                // fused-convolution order adds the fp32 noise to the half activation (one rounding); the un-fused order rounds the noise first
                if (a.round_noise || SCALE) {
                    const h2 n2 = {(_Float16)nv[k], (_Float16)nv[k + 1]};
                    r = __builtin_elementwise_fma(xv, s2, n2);
                } else {
                    r = (h2){(_Float16)((float)xv[0] + nv[k]), (_Float16)((float)xv[1] + nv[k + 1])};
                }
            } else {
                r = xv * s2;
            }
            t[k] = (float)r[0];
            t[k + 1] = (float)r[1];
        }
    } else {
        const float sc = SCALE ? round_to<T>(a.scale[row]) : 1.f;
#pragma unroll
        for (int k = 0; k < VEC; k++) {
            float u = float(load_as<T>(in.v, k));
            if constexpr (SCALE || NOISE) u = round_to<T>(fmaf(u, sc, NOISE ? ((a.round_noise || SCALE) ? round_to<T>(nv[k]) : nv[k]) : 0.f));
            t[k] = u;
        }
    }
    P out;
#pragma unroll
    for (int k = 0; k < VEC; k++) {
        const float u = t[k] + bv;
        float r = u;
        if (ACT == 3) r = (a.alpha >= 0.f && a.alpha <= 1.f) ? fmaxf(u, u * a.alpha) : (u > 0.f ? u : u * a.alpha);          // lrelu (see common.h)
        r *= a.gain;
        if (a.clamp >= 0.f) r = __builtin_amdgcn_fmed3f(r, -a.clamp, a.clamp);                     // NaN -> -clamp (bias_act.cu:143)
        store_as<T>(out.v, k, r);
    }
    *reinterpret_cast<P*>(y + v * VEC) = out;
}

template <class T, int VEC>
__global__ __launch_bounds__(kThreads) void scale_channels_kernel(const T* __restrict__ x, const float* __restrict__ s, T* __restrict__ y, unsigned row_len) {
    typedef Pk<T, VEC> P;
    const unsigned row = blockIdx.y;
    const unsigned v = blockIdx.x * kThreads + threadIdx.x;
    if (v * VEC >= row_len) return;
    const float sc = round_to<T>(s[row]);
    const P in = *reinterpret_cast<const P*>(x + int64_t(row) * row_len + v * VEC);
    P out;
#pragma unroll
    for (int k = 0; k < VEC; k++) store_as<T>(out.v, k, float(load_as<T>(in.v, k)) * sc);
    *reinterpret_cast<P*>(y + int64_t(row) * row_len + v * VEC) = out;
}

// ---------------------------------------------------------------------------------------------
// Channels_last forms (memory [N, pixels, C]).  The fp16 blocks of the generator run in the layout MIOpen's fp16 kernels compute
// in, which removes its NCHW<->NHWC transposes around every convolution (tools/bench_sr_conv_layout.py: 0.63 ms of the 3.8 ms
// superresolution at batch 4).  A lane owns one 16-byte vector of adjacent channels of one pixel; per-channel operands (bias,
// demodulation, the next layer's styles) are vectors too, per-pixel operands (noise) one scalar.

struct EpiNhwcArgs {
    const void* x; void* y;
    const float* scale;      // [n, channels] or NULL
    const float* noise;      // [pixels] (or [n, pixels] when noise_per_item) or NULL
    const void* bias;        // [channels], activations' dtype, or NULL
    const float* next_scale; // [n, channels] or NULL: the result is additionally multiplied by these (the next layer's input scaling)
    unsigned pixels, channels;
    int noise_per_item, round_noise, act;
    float alpha, gain, clamp;
};

// y[n, p, c] = round_T(clamp(act(round_T(x * T(scale[n,c]) + noise[p]) + bias[c]) * gain)) (* T(next_scale[n,c]), rounded again):
// the same roundings, in the same order, as modconv_epilogue_kernel followed by scale_channels_kernel.
// A lane keeps ONE channel vector and walks over pixels (FIXED: the workgroup's 256 lanes are 256 / cv pixel rows of cv channel
// vectors, cv = channels / VEC dividing 256), so bias, demodulation and next-layer scale are loaded once per lane, not once per
// element: the first version re-loaded them per vector (three extra vector loads next to the 16 bytes of data) and ran at a
// quarter of the memory rate.  Other channel counts take the per-vector form (FIXED = false).
template <class T, int VEC, int ACT, bool SCALE, bool NOISE, bool NEXT, bool FIXED>
__global__ __launch_bounds__(kThreads) void modconv_epilogue_nhwc_kernel(EpiNhwcArgs a, unsigned pixels_per_block) {
    typedef Pk<T, VEC> P;
    const unsigned n = blockIdx.y;
    const unsigned cv = a.channels / VEC;
    unsigned pix, pix_end, pix_step, c0;
    if constexpr (FIXED) {
        c0 = (threadIdx.x % cv) * VEC;
        pix_step = kThreads / cv;
        pix = blockIdx.x * pixels_per_block + threadIdx.x / cv;
        pix_end = min(a.pixels, (blockIdx.x + 1) * pixels_per_block);
    } else {
        const unsigned v = blockIdx.x * kThreads + threadIdx.x;
        if (v >= a.pixels * cv) return;
        pix = v / cv; c0 = (v % cv) * VEC;
        pix_end = pix + 1; pix_step = 1;
    }
    // the lane's channel vector of the per-channel operands: whole vectors under uniform conditions where the channel count allows
    // (element-by-element conditional loads compile to one dependent round trip each -- 24 of them in front of 16 data vectors)
    float sc[VEC], nx[VEC], bv[VEC];
#pragma unroll
    for (int k = 0; k < VEC; k++) { sc[k] = 1.f; nx[k] = 1.f; bv[k] = 0.f; }
    if constexpr (VEC % 4 == 0) {
        float4 sv[VEC / 4], nv[VEC / 4];
        P bb;
        const bool has_b = a.bias != nullptr;
        if constexpr (SCALE) {
#pragma unroll
            for (int q = 0; q < VEC / 4; q++) sv[q] = *reinterpret_cast<const float4*>(a.scale + int64_t(n) * a.channels + c0 + 4 * q);
        }
        if constexpr (NEXT) {
#pragma unroll
            for (int q = 0; q < VEC / 4; q++) nv[q] = *reinterpret_cast<const float4*>(a.next_scale + int64_t(n) * a.channels + c0 + 4 * q);
        }
        if (has_b) bb = *reinterpret_cast<const P*>(static_cast<const T*>(a.bias) + c0);
        if constexpr (SCALE) {
#pragma unroll
            for (int q = 0; q < VEC / 4; q++) { sc[4 * q] = sv[q].x; sc[4 * q + 1] = sv[q].y; sc[4 * q + 2] = sv[q].z; sc[4 * q + 3] = sv[q].w; }
        }
        if constexpr (NEXT) {
#pragma unroll
            for (int q = 0; q < VEC / 4; q++) {
                nx[4 * q] = round_to<T>(nv[q].x); nx[4 * q + 1] = round_to<T>(nv[q].y); nx[4 * q + 2] = round_to<T>(nv[q].z); nx[4 * q + 3] = round_to<T>(nv[q].w);
            }
        }
        if (has_b) {
#pragma unroll
            for (int k = 0; k < VEC; k++) bv[k] = float(load_as<T>(bb.v, k));
        }
    } else {
#pragma unroll
        for (int k = 0; k < VEC; k++) {
            sc[k] = SCALE ? a.scale[int64_t(n) * a.channels + c0 + k] : 1.f;
            nx[k] = NEXT ? round_to<T>(a.next_scale[int64_t(n) * a.channels + c0 + k]) : 1.f;
            bv[k] = a.bias ? float(load_as<T>(static_cast<const T*>(a.bias), c0 + k)) : 0.f;
        }
    }
    const T* xn = static_cast<const T*>(a.x) + int64_t(n) * a.pixels * a.channels + c0;
    T* yn = static_cast<T*>(a.y) + int64_t(n) * a.pixels * a.channels + c0;
    const float* nzp = NOISE ? a.noise + (a.noise_per_item ? int64_t(n) * a.pixels : 0) : nullptr;
    constexpr int UN = FIXED ? 4 : 1;
    for (; pix < pix_end; pix += pix_step * UN) {
        P in[UN];
        float nzs[UN];
#pragma unroll
        for (int u = 0; u < UN; u++) {
            const unsigned pu = min(pix + u * pix_step, a.pixels - 1);
            in[u] = *reinterpret_cast<const P*>(xn + int64_t(pu) * a.channels);
            nzs[u] = NOISE ? nzp[pu] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < UN; u++) {
            const unsigned pu = pix + u * pix_step;
            if (pu >= pix_end) break;
            const P out = modconv_epilogue_vec<T, VEC, ACT, SCALE, NOISE, NEXT>(in[u], sc, nzs[u], a.round_noise != 0, bv, nx, a.alpha, a.gain, a.clamp);
            *reinterpret_cast<P*>(yn + int64_t(pu) * a.channels) = out;
        }
    }
}

template <class T, int VEC>
__global__ __launch_bounds__(kThreads) void scale_channels_nhwc_kernel(const T* __restrict__ x, const float* __restrict__ s, T* __restrict__ y,
                                                                       unsigned pixels, unsigned channels) {
    typedef Pk<T, VEC> P;
    const unsigned n = blockIdx.y;
    const unsigned cv = channels / VEC;
    const unsigned v = blockIdx.x * kThreads + threadIdx.x;
    if (v >= pixels * cv) return;
    const unsigned c0 = (v % cv) * VEC;
    const int64_t base = int64_t(n) * pixels * channels + int64_t(v) * VEC;
    const P in = *reinterpret_cast<const P*>(x + base);
    P out;
#pragma unroll
    for (int k = 0; k < VEC; k++) store_as<T>(out.v, k, float(load_as<T>(in.v, k)) * round_to<T>(s[int64_t(n) * channels + c0 + k]));
    *reinterpret_cast<P*>(y + base) = out;
}

// ToRGBLayer on a channels_last fp16 tensor (networks_stylegan2.py:349-367 with the fused modulation of :89-96): a modulated 1x1
// convolution to THREE channels, bias and clamp -- y[n,o,p] = clamp(half(sum_c x[n,p,c] * half(weight[o,c] * styles[n,c])) + bias[o]).
// It is a streaming read of x: C/8 lanes share a pixel, one 16-byte vector each, three v_dot2_f32_f16 chains per lane on the packed
// halves as loaded, a butterfly over the pixel's lanes, the first three lanes store (y is NCHW: it is added to the running image).
// Replaces scale_channels (read + write of x) + MIOpen's 1x1 convolution + the bias/clamp pass: 0.22 -> 0.06 ms on [4,128,512,512].
// ACCUM: instead of storing y, add it to the running fp32 image `img` [n,3,pixels] in place -- the block's `img.add_(y.to(float32))`
// (networks_stylegan2.py:461-463) with the same roundings (y rounded to fp16 first), without materialising y: two launches fewer per block.
template <int LPP, bool ACCUM>                             // lanes per pixel = C / 8
__global__ __launch_bounds__(kThreads) void torgb_nhwc_kernel(const __half* __restrict__ x, const float* __restrict__ weight, const float* __restrict__ styles,
                                                              const __half* __restrict__ bias, __half* __restrict__ y, float* __restrict__ img,
                                                              unsigned pixels, float clamp) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    constexpr int C = LPP * 8, PPB = kThreads / LPP;      // pixels per workgroup step
    constexpr int UNROLL = 4;
    const unsigned n = blockIdx.y;
    const unsigned sub = threadIdx.x % LPP, grp = threadIdx.x / LPP;
    h2 w[3][4];
#pragma unroll
    for (int o = 0; o < 3; o++)
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const unsigned c = sub * 8 + 2 * k;
            w[o][k] = (h2){(_Float16)(weight[o * C + c] * styles[n * C + c]), (_Float16)(weight[o * C + c + 1] * styles[n * C + c + 1])};
        }
    const float b = sub < 3 && bias ? __half2float(bias[sub]) : 0.f;
    const __half* xn = x + int64_t(n) * pixels * C;
    __half* yn = ACCUM ? nullptr : y + int64_t(n) * 3 * pixels;
    float* in_ = ACCUM ? img + int64_t(n) * 3 * pixels : nullptr;
    for (unsigned p0 = (blockIdx.x * UNROLL) * PPB; p0 < pixels; p0 += gridDim.x * UNROLL * PPB) {
        uint4 raw[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            const unsigned p = min(p0 + u * PPB + grp, pixels - 1);
            raw[u] = *reinterpret_cast<const uint4*>(xn + int64_t(p) * C + sub * 8);
        }
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            const unsigned p = p0 + u * PPB + grp;
            const unsigned q[4] = {raw[u].x, raw[u].y, raw[u].z, raw[u].w};
            float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const h2 xv = __builtin_bit_cast(h2, q[k]);
#pragma unroll
                for (int o = 0; o < 3; o++) acc[o] = __builtin_amdgcn_fdot2(xv, w[o][k], acc[o], false);
            }
#pragma unroll
            for (int off = 1; off < LPP; off <<= 1)
#pragma unroll
                for (int o = 0; o < 3; o++) acc[o] += __shfl_xor(acc[o], off);
            if (sub < 3 && p < pixels) {
                float r = __half2float(__float2half(sub == 0 ? acc[0] : (sub == 1 ? acc[1] : acc[2]))) + b;
                if (clamp >= 0.f) r = __builtin_amdgcn_fmed3f(r, -clamp, clamp);          // NaN -> -clamp (bias_act.cu:143)
                if constexpr (ACCUM) in_[int64_t(sub) * pixels + p] += __half2float(__float2half(r));
                else                 yn[int64_t(sub) * pixels + p] = __float2half(r);
            }
        }
    }
}

}  // namespace

extern "C" int gnerf_modulate_weights(const float* weight, const float* styles, void* out, int out_dtype, float* dcoefs,
                                      int n, int n_out, int n_in, int kk, int demodulate, int prenorm, int out_layout, gnerf_stream_t stream) {
    using namespace gnerf;
    if (!weight || !styles) return fail(GNERF_E_ARG, "modulate_weights: null pointer");
    if (n < 1 || n_out < 1 || n_in < 1 || kk < 1) return fail(GNERF_E_ARG, "modulate_weights: empty shape");
    if (!out && !dcoefs) return GNERF_OK;
    if (dcoefs && !demodulate) return fail(GNERF_E_ARG, "modulate_weights: dcoefs requested without demodulation");
    if (out_layout < 0 || out_layout > 3) return fail(GNERF_E_ARG, "modulate_weights: out_layout must be GNERF_W_OIK .. GNERF_W_IKO");
    const dim3 g((unsigned)(n * n_out)), b(kThreads);
    if (!out || out_dtype == GNERF_F32)
        hipLaunchKernelGGL(modulate_weights_kernel<float>, g, b, 0, as_stream(stream), weight, styles, static_cast<float*>(out), dcoefs, n_out, n_in, kk, demodulate, prenorm, out_layout);
    else if (out_dtype == GNERF_F16)
        hipLaunchKernelGGL(modulate_weights_kernel<__half>, g, b, 0, as_stream(stream), weight, styles, static_cast<__half*>(out), dcoefs, n_out, n_in, kk, demodulate, prenorm, out_layout);
    else
        return fail(GNERF_E_ARG, "modulate_weights: output dtype must be float32 or float16");
    return check_launch("modulate_weights");
}

extern "C" int gnerf_normalise_styles(const float* styles, float* out, int n, int n_in, gnerf_stream_t stream) {
    using namespace gnerf;
    if (!styles || !out || n < 1 || n_in < 1) return fail(GNERF_E_ARG, "normalise_styles: bad argument");
    hipLaunchKernelGGL(normalise_styles_kernel, dim3((unsigned)n), dim3(kThreads), 0, as_stream(stream), styles, out, n_in);
    return check_launch("normalise_styles");
}

extern "C" int gnerf_scale_channels(const void* x, const float* scale, void* y, int dtype, int rows, int row_len, gnerf_stream_t stream) {
    using namespace gnerf;
    if (!x || !scale || !y) return fail(GNERF_E_ARG, "scale_channels: null pointer");
    if (rows < 1 || row_len < 1 || rows > 65535) return fail(GNERF_E_ARG, "scale_channels: rows must be in 1..65535");
    const bool al = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0;
    if (dtype == GNERF_F16) {
        if (al && row_len % 8 == 0) hipLaunchKernelGGL((scale_channels_kernel<__half, 8>), dim3((row_len / 8 + kThreads - 1) / kThreads, rows), dim3(kThreads), 0, as_stream(stream),
                                                       static_cast<const __half*>(x), scale, static_cast<__half*>(y), unsigned(row_len));
        else hipLaunchKernelGGL((scale_channels_kernel<__half, 1>), dim3((row_len + kThreads - 1) / kThreads, rows), dim3(kThreads), 0, as_stream(stream),
                                static_cast<const __half*>(x), scale, static_cast<__half*>(y), unsigned(row_len));
    } else if (dtype == GNERF_F32) {
        if (al && row_len % 4 == 0) hipLaunchKernelGGL((scale_channels_kernel<float, 4>), dim3((row_len / 4 + kThreads - 1) / kThreads, rows), dim3(kThreads), 0, as_stream(stream),
                                                       static_cast<const float*>(x), scale, static_cast<float*>(y), unsigned(row_len));
        else hipLaunchKernelGGL((scale_channels_kernel<float, 1>), dim3((row_len + kThreads - 1) / kThreads, rows), dim3(kThreads), 0, as_stream(stream),
                                static_cast<const float*>(x), scale, static_cast<float*>(y), unsigned(row_len));
    } else {
        return fail(GNERF_E_ARG, "scale_channels: dtype must be float32 or float16");
    }
    return check_launch("scale_channels");
}

extern "C" int gnerf_modconv_epilogue(const void* x, void* y, int dtype, int rows, int row_len, int channels,
                                      const float* scale, const float* noise, int noise_per_item, int round_noise, const void* bias,
                                      int act, float alpha, float gain, float clamp, gnerf_stream_t stream) {
    using namespace gnerf;
    if (!x || !y) return fail(GNERF_E_ARG, "modconv_epilogue: null pointer");
    if (rows < 1 || row_len < 1 || channels < 1 || rows % channels != 0 || rows > 65535) return fail(GNERF_E_ARG, "modconv_epilogue: bad shape (rows %d, channels %d)", rows, channels);
    if (act != 1 && act != 3) return fail(GNERF_E_UNSUPPORTED, "modconv_epilogue: only linear and lrelu");
    EpiArgs a{x, y, scale, noise, bias, unsigned(row_len), unsigned(channels), noise_per_item, round_noise, act, alpha, gain, clamp};
    const bool al = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0;
    hipStream_t s = as_stream(stream);
#define GNERF_EPI3(T_, V_, A_) do { \
        if (scale && noise)  hipLaunchKernelGGL((modconv_epilogue_kernel<T_, V_, A_, true, true>), g, dim3(kThreads), 0, s, a); \
        else if (scale)      hipLaunchKernelGGL((modconv_epilogue_kernel<T_, V_, A_, true, false>), g, dim3(kThreads), 0, s, a); \
        else if (noise)      hipLaunchKernelGGL((modconv_epilogue_kernel<T_, V_, A_, false, true>), g, dim3(kThreads), 0, s, a); \
        else                 hipLaunchKernelGGL((modconv_epilogue_kernel<T_, V_, A_, false, false>), g, dim3(kThreads), 0, s, a); } while (0)
#define GNERF_EPI(T_, V_) do { const dim3 g((row_len / V_ + kThreads - 1) / kThreads, rows); \
        if (act == 3) GNERF_EPI3(T_, V_, 3); else GNERF_EPI3(T_, V_, 1); } while (0)
    if (dtype == GNERF_F16) { if (al && row_len % 8 == 0) GNERF_EPI(__half, 8); else GNERF_EPI(__half, 1); }
    else if (dtype == GNERF_F32) { if (al && row_len % 4 == 0) GNERF_EPI(float, 4); else GNERF_EPI(float, 1); }
    else return fail(GNERF_E_ARG, "modconv_epilogue: dtype must be float32 or float16");
#undef GNERF_EPI
#undef GNERF_EPI3
    return check_launch("modconv_epilogue");
}

extern "C" int gnerf_scale_channels_nhwc(const void* x, const float* scale, void* y, int dtype, int n, int pixels, int channels, gnerf_stream_t stream) {
    using namespace gnerf;
    if (!x || !scale || !y) return fail(GNERF_E_ARG, "scale_channels_nhwc: null pointer");
    if (n < 1 || n > 65535 || pixels < 1 || channels < 1 || int64_t(pixels) * channels > INT32_MAX) return fail(GNERF_E_ARG, "scale_channels_nhwc: bad shape");
    const bool al = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0;
    hipStream_t s = as_stream(stream);
#define GNERF_SC(T_, V_) hipLaunchKernelGGL((scale_channels_nhwc_kernel<T_, V_>), dim3((unsigned)((int64_t(pixels) * (channels / V_) + kThreads - 1) / kThreads), n), dim3(kThreads), 0, s, \
                                            static_cast<const T_*>(x), scale, static_cast<T_*>(y), unsigned(pixels), unsigned(channels))
    if (dtype == GNERF_F16) { if (al && channels % 8 == 0) GNERF_SC(__half, 8); else GNERF_SC(__half, 1); }
    else if (dtype == GNERF_F32) { if (al && channels % 4 == 0) GNERF_SC(float, 4); else GNERF_SC(float, 1); }
    else return fail(GNERF_E_ARG, "scale_channels_nhwc: dtype must be float32 or float16");
#undef GNERF_SC
    return check_launch("scale_channels_nhwc");
}

extern "C" int gnerf_modconv_epilogue_nhwc(const void* x, void* y, int dtype, int n, int pixels, int channels,
                                           const float* scale, const float* noise, int noise_per_item, int round_noise, const void* bias,
                                           int act, float alpha, float gain, float clamp, const float* next_scale, gnerf_stream_t stream) {
    using namespace gnerf;
    if (!x || !y) return fail(GNERF_E_ARG, "modconv_epilogue_nhwc: null pointer");
    if (n < 1 || n > 65535 || pixels < 1 || channels < 1 || int64_t(pixels) * channels > INT32_MAX) return fail(GNERF_E_ARG, "modconv_epilogue_nhwc: bad shape");
    if (act != 1 && act != 3) return fail(GNERF_E_UNSUPPORTED, "modconv_epilogue_nhwc: only linear and lrelu");
    EpiNhwcArgs a{x, y, scale, noise, bias, next_scale, unsigned(pixels), unsigned(channels), noise_per_item, round_noise, act, alpha, gain, clamp};
    // the vector form also fetches the per-channel operands as 16-byte vectors
    const bool al = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(scale) |
                      reinterpret_cast<uintptr_t>(next_scale) | reinterpret_cast<uintptr_t>(bias)) & 15) == 0;
    hipStream_t s = as_stream(stream);
#define GNERF_EPI5(T_, V_, A_, S_, N_, X_) do { \
        if (fixed) hipLaunchKernelGGL((modconv_epilogue_nhwc_kernel<T_, V_, A_, S_, N_, X_, true>), g, dim3(kThreads), 0, s, a, ppb); \
        else       hipLaunchKernelGGL((modconv_epilogue_nhwc_kernel<T_, V_, A_, S_, N_, X_, false>), g, dim3(kThreads), 0, s, a, ppb); } while (0)
#define GNERF_EPI4(T_, V_, A_, S_, N_) do { if (next_scale) GNERF_EPI5(T_, V_, A_, S_, N_, true); else GNERF_EPI5(T_, V_, A_, S_, N_, false); } while (0)
#define GNERF_EPI3(T_, V_, A_) do { \
        if (scale && noise)  GNERF_EPI4(T_, V_, A_, true, true); \
        else if (scale)      GNERF_EPI4(T_, V_, A_, true, false); \
        else if (noise)      GNERF_EPI4(T_, V_, A_, false, true); \
        else                 GNERF_EPI4(T_, V_, A_, false, false); } while (0)
#define GNERF_EPI(T_, V_) do { const unsigned cv = channels / V_; const bool fixed = cv <= kThreads && kThreads % cv == 0; \
        const unsigned ppb = fixed ? (kThreads / cv) * 16 : 0; \
        const dim3 g(fixed ? (unsigned)((pixels + ppb - 1) / ppb) : (unsigned)((int64_t(pixels) * cv + kThreads - 1) / kThreads), n); \
        if (act == 3) GNERF_EPI3(T_, V_, 3); else GNERF_EPI3(T_, V_, 1); } while (0)
    if (dtype == GNERF_F16) { if (al && channels % 8 == 0) GNERF_EPI(__half, 8); else GNERF_EPI(__half, 1); }
    else if (dtype == GNERF_F32) { if (al && channels % 4 == 0) GNERF_EPI(float, 4); else GNERF_EPI(float, 1); }
    else return fail(GNERF_E_ARG, "modconv_epilogue_nhwc: dtype must be float32 or float16");
#undef GNERF_EPI
#undef GNERF_EPI3
#undef GNERF_EPI4
#undef GNERF_EPI5
    return check_launch("modconv_epilogue_nhwc");
}

static int torgb_nhwc_impl(const void* x, const float* weight, const float* styles, const void* bias, void* y, float* img,
                           int n, int pixels, int channels, float clamp, gnerf_stream_t stream) {
    using namespace gnerf;
    if (!x || !weight || !styles || (!y && !img)) return fail(GNERF_E_ARG, "torgb_nhwc: null pointer");
    if (n < 1 || n > 65535 || pixels < 1) return fail(GNERF_E_ARG, "torgb_nhwc: bad shape");
    if (reinterpret_cast<uintptr_t>(x) & 15) return fail(GNERF_E_ARG, "torgb_nhwc: x must be 16-byte aligned");
    hipStream_t s = as_stream(stream);
    const __half* xh = static_cast<const __half*>(x);
    const __half* bh = static_cast<const __half*>(bias);
    __half* yh = static_cast<__half*>(y);
#define GNERF_RGB(L_) do { const int ppb = kThreads / L_ * 4; int blocks = (pixels + ppb - 1) / ppb; if (blocks > kNumCU * 8) blocks = kNumCU * 8; \
        if (img) hipLaunchKernelGGL((torgb_nhwc_kernel<L_, true>), dim3((unsigned)blocks, n), dim3(kThreads), 0, s, xh, weight, styles, bh, yh, img, unsigned(pixels), clamp); \
        else     hipLaunchKernelGGL((torgb_nhwc_kernel<L_, false>), dim3((unsigned)blocks, n), dim3(kThreads), 0, s, xh, weight, styles, bh, yh, img, unsigned(pixels), clamp); } while (0)
    switch (channels) {
        case 32: GNERF_RGB(4); break;
        case 64: GNERF_RGB(8); break;
        case 128: GNERF_RGB(16); break;
        case 256: GNERF_RGB(32); break;
        case 512: GNERF_RGB(64); break;
        default: return fail(GNERF_E_UNSUPPORTED, "torgb_nhwc: %d input channels (32, 64, 128, 256 or 512)", channels);
    }
#undef GNERF_RGB
    return check_launch("torgb_nhwc");
}

extern "C" int gnerf_torgb_nhwc(const void* x, const float* weight, const float* styles, const void* bias, void* y,
                                int n, int pixels, int channels, float clamp, gnerf_stream_t stream) {
    if (!y) return gnerf::fail(GNERF_E_ARG, "torgb_nhwc: null pointer");
    return torgb_nhwc_impl(x, weight, styles, bias, y, nullptr, n, pixels, channels, clamp, stream);
}

extern "C" int gnerf_torgb_nhwc_accumulate(const void* x, const float* weight, const float* styles, const void* bias, float* img,
                                           int n, int pixels, int channels, float clamp, gnerf_stream_t stream) {
    if (!img) return gnerf::fail(GNERF_E_ARG, "torgb_nhwc_accumulate: null pointer");
    return torgb_nhwc_impl(x, weight, styles, bias, nullptr, img, n, pixels, channels, clamp, stream);
}
