/*
 * gnerf_hip.h -- C ABI of libgnerf_hip.so, the MI355X (gfx950) implementation of G-NeRF's
 * hot path: the tri-plane importance renderer and the StyleGAN custom ops.
 *
 * Plain pointers and sizes only; every pointer is a DEVICE pointer unless it says "host".
 * All entry points enqueue work on `stream` (a hipStream_t passed as void*) and return
 * immediately; they return 0 on success and a negative GNERF_E_* code on failure
 * (gnerf_last_error() then holds a message for the calling thread).  No entry point
 * allocates, frees or synchronises, so all of them are hipGraph-capturable.
 *
 * Reference interfaces replaced (paths relative to the reference's g_nerf/):
 *   gnerf_bias_act          <- bias_act_plugin.bias_act            torch_utils/ops/bias_act.cpp:36
 *   gnerf_upfirdn2d         <- upfirdn2d_plugin.upfirdn2d          torch_utils/ops/upfirdn2d.cpp:20
 *   gnerf_filtered_lrelu    <- filtered_lrelu_plugin.filtered_lrelu      torch_utils/ops/filtered_lrelu.cpp:20
 *   gnerf_filtered_lrelu_act<- filtered_lrelu_plugin.filtered_lrelu_act_ torch_utils/ops/filtered_lrelu.cpp:217
 *   gnerf_grid_sample_2d(_backward) <- torch_utils/ops/grid_sample_gradfix.py:45, :62-77 (ATen's sampler upstream)
 *   gnerf_modulate_weights / gnerf_scale_channels / gnerf_modconv_epilogue <- the ATen elementwise chains of modulated_conv2d
 *                              and SynthesisLayer.forward                training/networks_stylegan2.py:41-98, :315-334
 *   gnerf_render_forward    <- ImportanceRenderer.forward          training/volumetric_rendering/renderer.py:88-140
 *                              (pure PyTorch in the reference; there is no native counterpart)
 *   gnerf_query_points      <- ImportanceRenderer.run_model        training/volumetric_rendering/renderer.py:142-148
 *   gnerf_make_rays         <- RaySampler.forward                  training/volumetric_rendering/ray_sampler.py:24-63
 *   gnerf_planes_to_nhwc    <- (layout change feeding the renderer; the reference keeps NCHW, triplane.py:74)
 *   gnerf_planes_to_nhwc_stats / gnerf_planes_absmax <- (the same + max |planes|, which picks the decoder arithmetic)
 *   gnerf_upsample2x_add_nhwc <- the backbone's last `upsample2d(img) + torgb(x)` (networks_stylegan2.py:456-463), channels_last
 *   gnerf_planes_from_nhwc  <- (the same for the plane gradient on the way back)
 *   gnerf_render_backward   <- autograd through renderer.py:88-140 (grid_sample_gradfix.py:62-77 for the planes)
 */
#ifndef GNERF_HIP_H
#define GNERF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GNERF_ABI_VERSION 11

/* error codes */
#define GNERF_OK            0
#define GNERF_E_ARG        -1   /* invalid argument (shape, size, null pointer, unsupported value) */
#define GNERF_E_LAUNCH     -2   /* HIP reported an error at launch */
#define GNERF_E_UNSUPPORTED -3  /* valid request, but no kernel for it (caller may use a generic path) */

/* element types of activation tensors */
#define GNERF_F32 0
#define GNERF_F16 1
#define GNERF_F64 2

typedef void* gnerf_stream_t;   /* hipStream_t */

int         gnerf_abi_version(void);
const char* gnerf_last_error(void);          /* thread-local, host pointer */
/* (ABI 9) Measurement aid: one wave samples the shader-cycle counter against the 100 MHz reference counter for `microseconds` and writes
 * out[0] = shader cycles, out[1] = reference ticks (device memory, 16 bytes): clock = out[0] / out[1] x 100 MHz.  Launch it on a side
 * stream while the kernels of interest run: the chip's clock under THAT load (bench.py's roofline quotes it). */
int gnerf_clock_sample(unsigned long long* out, double microseconds, gnerf_stream_t stream);
const char* gnerf_build_info(void);          /* e.g. "gfx950 hipcc ..." , host pointer */

/* ------------------------------------------------------------------------------------------
 * bias_act.  y = clamp(act(x + b) * gain)  and its first / second derivative forms.
 * Same contract as bias_act.cpp:36-94: x dense with `numel` elements in any memory format;
 * b has size_b elements and element i of x uses b[(i / step_b) % size_b] (step_b = stride of
 * the bias dimension); absent tensors are NULL.  grad: 0 forward (x = input),
 * 1 first order (x = dy, needs xref or yref as the activation dictates),
 * 2 second order (x = d_dx, dy = the first-order incoming gradient).
 * act: 1 linear 2 relu 3 lrelu 4 tanh 5 sigmoid 6 elu 7 selu 8 softplus 9 swish.
 * clamp < 0 disables clamping.
 * NaN: with clamp >= 0 a NaN result is returned as -clamp (the clamp is one v_med3_f32; the reference's kernel does the same,
 * bias_act.cu:143), in every native epilogue (gnerf_bias_act, gnerf_modconv_epilogue*, gnerf_torgb_nhwc*, gnerf_blur4_epilogue_nhwc); the
 * PyTorch-op forms that CPU tensors take keep the NaN. */
int gnerf_bias_act(const void* x, const void* b, const void* xref, const void* yref, const void* dy,
                   void* y, int dtype, int64_t numel, int size_b, int64_t step_b,
                   int grad, int act, float alpha, float gain, float clamp, gnerf_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * The surroundings of StyleGAN2's modulated convolution (networks_stylegan2.py:41-98 modulated_conv2d, :315-334
 * SynthesisLayer.forward; the convolution itself stays MIOpen's).  SURVEY.md section 8f.3.
 *
 * gnerf_modulate_weights: out[n,o,i,k] = weight[o,i,k] * styles[n,i] * dcoef[n,o] with dcoef = rsqrt(sum_ik (weight styles)^2 + 1e-8)
 * when `demodulate` (:66-75), after the fp16 pre-normalisation of :62-64 when `prenorm` (weight[o] / (sqrt(n_in*kk) max|weight[o]|),
 * styles[n] / max|styles[n]|).  weight [n_out, n_in, kk] and styles [n, n_in] float32; out [n, n_out, n_in, kk] in out_dtype
 * (GNERF_F32 / GNERF_F16) or NULL; dcoefs [n, n_out] float32 or NULL (the un-fused form needs only these, :71-72).
 * out_layout: memory order of each sample's weights -- GNERF_W_OIK [n_out, n_in, kk] (conv2d), GNERF_W_IOK [n_in, n_out, kk]
 * (what conv_transpose2d takes: the x2-upsampling layers, conv2d_resample.py:114-123, otherwise a strided copy of all weights
 * per call), GNERF_W_OKI / GNERF_W_IKO: the channels_last memory of those two ([O,k,k,I] / [I,k,k,O]). */
#define GNERF_W_OIK 0
#define GNERF_W_IOK 1
#define GNERF_W_OKI 2
#define GNERF_W_IKO 3
int gnerf_modulate_weights(const float* weight, const float* styles, void* out, int out_dtype, float* dcoefs,
                           int n, int n_out, int n_in, int kk, int demodulate, int prenorm, int out_layout, gnerf_stream_t stream);
/* styles[n,:] / max|styles[n,:]| (the pre-normalised styles the un-fused form scales the activations with, :64 and :77). */
int gnerf_normalise_styles(const float* styles, float* out, int n, int n_in, gnerf_stream_t stream);
/* y[r, :] = x[r, :] * scale[r] for r < rows (rows = batch * channels of an NCHW tensor, row_len = H*W); the product is formed in
 * the activations' dtype like `x * styles.to(x.dtype)` (:77). */
int gnerf_scale_channels(const void* x, const float* scale, void* y, int dtype, int rows, int row_len, gnerf_stream_t stream);
/* One pass for everything after the convolution:
 *   t = x * scale[row] + noise            (un-fused demodulation + noise, fma.fma at :79-80; or `x.add_(noise)` at :96-97; rounded to
 *                                          the activations' dtype as the materialised tensor would be; skipped when both are NULL)
 *   y = clamp(act(t + bias[row % channels]) * gain)                                          (bias_act, :331-333)
 * x, y: [rows, row_len] NCHW activations (rows = batch * channels <= 65535); scale: float32 [rows] or NULL; noise: float32
 * [row_len] (or [batch, row_len] when noise_per_item: 'random' mode) or NULL, rounded to the activations' dtype first when
 * round_noise or when scale is given (the un-fused form's fma takes a noise tensor of the activations' dtype, :79-80); bias: [channels] in the activations' dtype or NULL; act: 1 linear or 3 lrelu (GNERF_E_UNSUPPORTED otherwise). */
int gnerf_modconv_epilogue(const void* x, void* y, int dtype, int rows, int row_len, int channels,
                           const float* scale, const float* noise, int noise_per_item, int round_noise, const void* bias,
                           int act, float alpha, float gain, float clamp, gnerf_stream_t stream);
/* The same three steps for CHANNELS_LAST activations (memory [n, pixels, channels]; the layout the fp16 blocks run in so that
 * MIOpen's fp16 convolutions need no transposes, and what the reference's `fp16_channels_last` produces, networks_stylegan2.py:385).
 * scale / next_scale: float32 [n, channels] or NULL; noise: float32 [pixels] (or [n, pixels] when noise_per_item) or NULL.
 * next_scale: the result, rounded to the activations' dtype, is additionally multiplied by next_scale[n, c] -- the NEXT layer's
 * `x * styles` (:77) folded into this pass (bit-identical to gnerf_modconv_epilogue_nhwc followed by gnerf_scale_channels_nhwc). */
int gnerf_scale_channels_nhwc(const void* x, const float* scale, void* y, int dtype, int n, int pixels, int channels, gnerf_stream_t stream);
int gnerf_modconv_epilogue_nhwc(const void* x, void* y, int dtype, int n, int pixels, int channels,
                                const float* scale, const float* noise, int noise_per_item, int round_noise, const void* bias,
                                int act, float alpha, float gain, float clamp, const float* next_scale, gnerf_stream_t stream);
/* The 4x4 blur that follows a x2-upsampling transposed convolution (conv2d_resample.py:114-131) and the epilogue above in ONE pass over
 * channels_last activations: y = epilogue(round_T(blur(x) * blur_gain)), bit-identical to gnerf_upfirdn2d followed by
 * gnerf_modconv_epilogue_nhwc (no noise term: the layers that add noise run in NCHW float32 here).  x: [n, in_h, in_w, c] dense,
 * y: [n, out_h, out_w, c] dense, both 16-byte aligned with c filling whole 16-byte vectors; f: float32 [4,4] dense; padx0 / pady0 / flip
 * as in gnerf_upfirdn2d (up = down = 1); scale / next_scale: float32 [n, c] or NULL; bias: [c] in the activations' dtype or NULL. */
int gnerf_blur4_epilogue_nhwc(const void* x, const float* f, void* y, int dtype, int n, int c, int in_h, int in_w, int out_h, int out_w,
                              int padx0, int pady0, int flip, float blur_gain,
                              const float* scale, const void* bias, int act, float alpha, float gain, float clamp, const float* next_scale,
                              gnerf_stream_t stream);
/* (ABI 9) The 3x3 convolution of a modulated-convolution layer AND its epilogue in one launch, for float16 channels_last activations
 * in the shared-weight form (networks_stylegan2.py:41-98 as called from :315-334; the superresolution's 128 -> 128 @ 512^2 and
 * 256 -> 256 @ 256^2 layers, superresolution.py:285-303):
 *   y[n, p, o] = epilogue( half( sum_{ky,kx,c} w[o, c, ky, kx] * x[n, p + (ky - 1, kx - 1), c] ) )      zero padding, stride 1
 * with the epilogue of gnerf_modconv_epilogue_nhwc (act = lrelu; the convolution's fp32 accumulator is rounded to float16 first, as
 * a stand-alone convolution would have stored it, so the roundings match that function applied to the convolution's output).
 * x: [n, h, w, cin] float16; w_packed: [9, cout, cin] float16, tap-major (tap = ky * 3 + kx of the correlation form torch's conv2d
 * computes: w_packed[t, o, c] = weight[o, c, t / 3, t % 3]); y: [n, h, w, cout] float16; scale / next_scale: float32 [n, cout] or
 * NULL; noise: float32 [h * w] or NULL; bias: float16 [cout] or NULL; clamp < 0: none.  All 16-byte aligned.
 * w_packed's input-channel axis is cin ROUNDED UP to a multiple of 64, zero-filled (the kernel reads weights in 64-channel chunks and
 * masks the activations' missing channels).  GNERF_E_UNSUPPORTED unless h % 8 == 0, w % 32 == 0, cin % 8 == 0, cout % 128 == 0 (the caller
 * runs the two-launch form). */
int gnerf_conv3x3_epilogue_nhwc(const void* x, const void* w_packed, void* y, int n, int h, int w, int cin, int cout,
                                const float* scale, const float* noise, int round_noise, const void* bias,
                                float alpha, float gain, float clamp, const float* next_scale, gnerf_stream_t stream);
/* (ABI 11) The convolution of a block's LAST layer with the block's ToRGB taken in its epilogue -- SynthesisBlock.forward's
 * `x = conv1(x); y = torgb(x); img = img.add_(y)` (networks_stylegan2.py:452-463) where nothing else reads x (the superresolution's final block,
 * superresolution.py:285-303) -- as ONE launch that writes no x at all:
 *   img[n, o, p] += half( clamp( half( sum_c half(layer(x))[n, p, c] * rgb_w[n, o, c] ) + rgb_bias[o] ) )
 * i.e. gnerf_conv3x3_epilogue_nhwc (cout = 128, next_scale = NULL, a demodulation scale given) followed by gnerf_torgb_nhwc_accumulate, with the
 * same roundings: the layer's result is rounded to float16 where it would have been stored, the 1 x 1 products are v_dot2_f32_f16 on it.
 * rgb_w: float16 [n, 3, 128] = half(ToRGB weight[o, c] * ToRGB styles[n, c]); rgb_bias: float32 [3] (the float16 bias's values) or NULL;
 * rgb_clamp < 0: none; img: float32 [n, 3, h, w], dense, accumulated in place.  Other arguments and shape rules as gnerf_conv3x3_epilogue_nhwc. */
int gnerf_conv3x3_epilogue_torgb_nhwc(const void* x, const void* w_packed, int n, int h, int w, int cin,
                                      const float* scale, const float* noise, int round_noise, const void* bias,
                                      float alpha, float gain, float clamp,
                                      const void* rgb_w, const float* rgb_bias, float rgb_clamp, float* img, gnerf_stream_t stream);
/* (ABI 9) The stride-2 transposed 3x3 convolution of the x2 layers -- conv_transpose2d(x, w, stride = 2), what conv2d_resample.py:109-119
 * hands to the framework for up = 2 -- for float16 channels_last activations, as its four output phases on the matrix cores (the kernel
 * of gnerf_conv3x3_epilogue_nhwc with 4 / 2 / 2 / 1 of its taps; fp32 accumulation, the result rounded to float16 once).
 * x: [n, h, w, cin]; w_phases: [9, cout, cin rounded up to a multiple of 64, zero-filled] float16, the taps grouped by output phase (py, px) = (oy & 1, ox & 1):
 * (ky, kx) = (0,0), (0,2), (2,0), (2,2) | (0,1), (2,1) | (1,0), (1,2) | (1,1), with w_phases[t, o, c] = weight[c, o, ky, kx] of the transposed
 * convolution; y: [n, 2h + 1, 2w + 1, cout].  All 16-byte aligned.  GNERF_E_UNSUPPORTED unless cin % 8 == 0 and cout % 128 == 0. */
int gnerf_conv_transpose3x3_s2_nhwc(const void* x, const void* w_phases, void* y, int n, int h, int w, int cin, int cout, gnerf_stream_t stream);
/* (ABI 10) The same two convolutions in fp32-GRADE arithmetic, for the backbone's float32 layers (networks_stylegan2.py:41-98 and
 * conv2d_resample.py:48-143 on float32 activations; the reference hands them to the framework's fp32 convolution): every product x w is
 * evaluated as hi(x) hi(w) + lo(x) hi(w) + hi(x) lo(w) with hi(v) = half(v), lo(v) = half(v - hi(v)) on the f16 matrix instruction with
 * float32 accumulation -- the arithmetic of the fused renderer's decoder (dropped term and split roundings ~2^-21 relative, i.e. float32-
 * grade as long as the operands stay inside float16's RANGE: |x| < 65504) --, which for a convolution is the float16 kernel on three times
 * the input channels.
 *   gnerf_split_f16x3_nhwc: x float32 [n, pixels, channels] (channels_last), scale float32 [n, channels] or NULL -> y float16
 *     [n, pixels, 3 channels] = [hi | lo | hi] of v = x * scale (the layer's `x * styles`, :77, folded in).  channels % 8 == 0, 16-byte
 *     aligned.  |v| >= 65504 saturates to the largest finite half and sets *overflow (a device int the caller zeroed; NULL: not reported)
 *     to 1 -- the result of a convolution on such an operand is finite but wrong, and a caller that cannot bound its activations checks it.
 *   gnerf_conv3x3_f32x3_epilogue_nhwc: x3 as above with cin3 = 3 channels; w3_packed float16 [9, cout, cin3 rounded up to a multiple of 64,
 *     zero-filled] = [hi(w) | hi(w) | lo(w)] along the input-channel axis, tap-major like w_packed; bias float32 [cout] or NULL; y float32
 *     [n, h, w, cout]; epilogue operands and shape rules as gnerf_conv3x3_epilogue_nhwc (the epilogue runs in float32, nothing is rounded).
 *   gnerf_conv_transpose3x3_s2_f32x3_nhwc: likewise for the stride-2 transposed form, w3_phases in gnerf_conv_transpose3x3_s2_nhwc's tap
 *     order; y float32 [n, 2h + 1, 2w + 1, cout]. */
int gnerf_split_f16x3_nhwc(const float* x, const float* scale, void* y, int n, int pixels, int channels, int* overflow, gnerf_stream_t stream);
int gnerf_conv3x3_f32x3_epilogue_nhwc(const void* x3, const void* w3_packed, float* y, int n, int h, int w, int cin3, int cout,
                                      const float* scale, const float* noise, const float* bias,
                                      float alpha, float gain, float clamp, const float* next_scale, gnerf_stream_t stream);
int gnerf_conv_transpose3x3_s2_f32x3_nhwc(const void* x3, const void* w3_phases, float* y, int n, int h, int w, int cin3, int cout, gnerf_stream_t stream);
/* ToRGBLayer with three output channels on a channels_last float16 tensor (networks_stylegan2.py:349-367, modulation as in the
 * fused form :89-96): y[n, o, p] = clamp(half(sum_c x[n, p, c] * half(weight[o, c] * styles[n, c])) + bias[o]), products exact, fp32
 * accumulation.  x: float16 [n, pixels, channels] (channels 32, 64, 128, 256 or 512, 16-byte aligned); weight float32 [3, channels];
 * styles float32 [n, channels] (weight_gain already applied, :364); bias float16 [3] or NULL; y: float16 [n, 3, pixels] (NCHW: it is
 * added to the running image); clamp < 0 = none.  One streaming read of x instead of scale pass + 1x1 convolution + bias pass. */
int gnerf_torgb_nhwc(const void* x, const float* weight, const float* styles, const void* bias, void* y,
                     int n, int pixels, int channels, float clamp, gnerf_stream_t stream);
/* (ABI 7) The same layer ADDED to the block's running image instead of stored: img[n, o, p] += float(y[n, o, p]) with y as above
 * (rounded to float16 first), img float32 [n, 3, pixels] dense -- `img.add_(y.to(torch.float32))` of the 'skip' architecture
 * (networks_stylegan2.py:461-463) without materialising y or converting it: two launches fewer per block. */
int gnerf_torgb_nhwc_accumulate(const void* x, const float* weight, const float* styles, const void* bias, float* img,
                                int n, int pixels, int channels, float clamp, gnerf_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * upfirdn2d.  Zero-insert upsample, pad/crop, 2-D FIR, decimate (upfirdn2d.cpp:20-102).
 * x: [n, c, in_h, in_w] with element strides xs[4] = {stride_n, stride_c, stride_h, stride_w}
 * (NCHW and channels_last both allowed); y likewise with ys[4]; f: float32 [fh, fw] with
 * element strides fs[2] = {stride_h, stride_w}.
 * out_h = (in_h*upy + pady0 + pady1 - fh + downy) / downy, out_w likewise; the caller
 * allocates y with that shape (only pad*0 is needed by the kernel). */
int gnerf_upfirdn2d(const void* x, const float* f, void* y, int dtype,
                    int n, int c, int in_h, int in_w, const int64_t xs[4],
                    int fh, int fw, const int64_t fs[2],
                    int out_h, int out_w, const int64_t ys[4],
                    int upx, int upy, int downx, int downy, int padx0, int pady0,
                    int flip, float gain, gnerf_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * filtered_lrelu_act_: in-place gain * lrelu * clamp on x [n,c,h,w] (strides xs) with the
 * bit-packed 2-bit sign tensor (1 = negative, 2 = clamped; 4 pixels per byte, row pitch
 * s_w/4 bytes, s_w a multiple of 16) -- filtered_lrelu.cpp:217-294.
 * mode: 0 no signs, 1 write signs to s, 2 read signs from s (then x *= per-sign factor).
 * sx, sy: offset of x inside the sign tensor. */
int gnerf_filtered_lrelu_act(void* x, uint8_t* s, int dtype, int n, int c, int h, int w,
                             const int64_t xs[4], int s_h, int s_w, int sx, int sy,
                             float gain, float slope, float clamp, int mode, gnerf_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * filtered_lrelu, fused (filtered_lrelu.cpp:20-213): y = downfir(clamp(lrelu(upfir(x + b) * up^2 * gain))) in one
 * launch.  x: [n, c, xh, xw] float16/float32 with element strides xs[4]; b: [c], same dtype, contiguous;
 * fu, fd: float32 filters with fu_taps / fd_taps taps per axis, contiguous; *_rank 1 = separable (applied along
 * both axes), 2 = a 1x1 rank-2 filter (applied once).  y: [n, c, yh, yw] (strides ys) allocated by the caller with
 * yw = (xw*up + px0 + px1 - (fu_taps-1) - (fd_taps-1) + down-1) / down (only px0 / py0 are needed here).
 * Signs: s is the bit-packed tensor of gnerf_filtered_lrelu_act ([n, c, s_h, s_w/4] bytes), sign_mode 0 none,
 * 1 write (s_h = yh*down-(down-1)+fd_taps-1 rows, forward pass), 2 read (gradient pass: slope / zero factors come
 * from s at offset (sx, sy), values outside s pass through with the gain only).
 * Returns GNERF_E_UNSUPPORTED when no fused kernel covers the configuration (resampling factors other than
 * 1/2/4, more than 8 taps per polyphase branch, non-separable filters, float64): the caller then composes
 * gnerf_upfirdn2d + gnerf_filtered_lrelu_act + gnerf_upfirdn2d, which is what the reference does on return code -1
 * (filtered_lrelu.cpp:55-60, filtered_lrelu.py:225-231). */
int gnerf_filtered_lrelu(const void* x, const float* fu, const float* fd, const void* b, uint8_t* s, void* y,
                         int dtype, int n, int c, int xh, int xw, const int64_t xs[4],
                         int yh, int yw, const int64_t ys[4],
                         int fu_taps, int fu_rank, int fd_taps, int fd_rank,
                         int up, int down, int px0, int py0,
                         int s_h, int s_w, int sx, int sy, int sign_mode,
                         float gain, float slope, float clamp, int flip, gnerf_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Bilinear 2-D grid sampling (mode bilinear, zero padding, align_corners = False): what grid_sample_gradfix.grid_sample
 * evaluates upstream through torch.nn.functional.grid_sample (grid_sample_gradfix.py:45) and, in the backward,
 * aten::grid_sampler_2d_backward (grid_sample_gradfix.py:62-77).
 * image: [n, c, h, w] float16/float32 with element strides image_strides[4]; grid: [n, ho, wo, 2] float32 contiguous,
 * (x, y) in [-1, 1]; out: [n, c, ho, wo] contiguous, image's dtype. */
int gnerf_grid_sample_2d(const void* image, const float* grid, void* out, int dtype,
                         int n, int c, int h, int w, const int64_t image_strides[4], int ho, int wo, gnerf_stream_t stream);
/* The adjoint.  grad_out: [n, c, ho, wo] contiguous, image's dtype.  grad_image: float32 [n, c, h, w] contiguous or NULL;
 * grad_grid: float32 [n, ho, wo, 2] or NULL (needs image).  Both are ACCUMULATED into (the caller zeroes them). */
int gnerf_grid_sample_2d_backward(const void* grad_out, const void* image, const float* grid, float* grad_image, float* grad_grid,
                                  int dtype, int n, int c, int h, int w, const int64_t image_strides[4], int ho, int wo,
                                  gnerf_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Tri-plane layout change: NCHW float32 [np, c, h, w] -> NHWC [np, h, w, c] (np = 3*batch).
 * The renderer reads whole 128-byte texels (32 channels) from the NHWC copy. */
int gnerf_planes_to_nhwc(const float* planes_nchw, float* planes_nhwc, int np, int c, int h, int w,
                         gnerf_stream_t stream);
/* The same layout change, also measuring max |planes| on the way (the planes are read anyway): *absmax (one device
 * float, overwritten) feeds gnerf_render_params.planes_absmax.  A NaN anywhere in the planes makes *absmax NaN. */
int gnerf_planes_to_nhwc_stats(const float* planes_nchw, float* planes_nhwc, int np, int c, int h, int w,
                               float* absmax, gnerf_stream_t stream);
/* max |x| over `numel` floats (any layout) -> *absmax (one device float, overwritten). */
int gnerf_planes_absmax(const float* planes, int64_t numel, float* absmax, gnerf_stream_t stream);
/* The tri-plane producer's last step, writing the renderer's layout directly (no layout change afterwards):
 *   out = upfirdn2d(img, f, up=2, padding=[2,1,2,1], gain) + y          (networks_stylegan2.py:456-463, upfirdn2d.py:315-350)
 * img [n, c, h, w] float32 NCHW contiguous; y [n, c, 2h, 2w] float32 NCHW contiguous or NULL; f_host: the 4x4 filter, 16 floats
 * in HOST memory (row-major; flipped inside unless `flip`, like gnerf_upfirdn2d); out: [n, 2h, 2w, c] = channels_last memory of
 * [n, c, 2h, 2w], i.e. for c = 96 the interleaved plane layout (gnerf_render_params.planes_interleaved = 1).
 * absmax: optional device float receiving max |out| (gnerf_render_params.planes_absmax).
 * Returns GNERF_E_UNSUPPORTED unless c % 32 == 0, w % 16 == 0 and h % 2 == 0. */
int gnerf_upsample2x_add_nhwc(const float* img, const float* y, const float* f_host, int flip, float gain, float* out,
                              int n, int c, int h, int w, float* absmax, gnerf_stream_t stream);
/* The inverse, NHWC [np, h, w, c] -> NCHW [np, c, h, w]: hands gnerf_render_backward's plane gradient back in the layout
 * of the reference's planes (triplane.py:74). */
int gnerf_planes_from_nhwc(const float* planes_nhwc, float* planes_nchw, int np, int c, int h, int w,
                           gnerf_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Ray generation (ray_sampler.py:24-63): cam2world [n,4,4], intrinsics [n,3,3] row-major ->
 * origins, dirs [n, res*res, 3]; ray m = row*res + col looks through pixel centre
 * ((col+.5)/res, (row+.5)/res). */
/* Frame conversion of the video writer (gen_videos.py:173 and the permute in front of it): out[n, y, x, ch] =
 * uint8(clamp(img[n, ch, y, x] * 127.5 + 128, 0, 255)), product and sum rounded separately, truncating cast, NaN -> 0.
 * img float32 [n, c, h, w] dense, out uint8 [n, h, w, c] dense, 1 <= c <= 64.  One launch instead of four elementwise passes. */
int gnerf_to_uint8_nhwc(const float* img, unsigned char* out, int n, int c, int h, int w, gnerf_stream_t stream);

int gnerf_make_rays(const float* cam2world, const float* intrinsics, int n, int res,
                    float* origins, float* dirs, gnerf_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * The fused renderer (renderer.py:88-140 with MipRayMarcher2, ray_marcher.py:25-57, and the
 * OSGDecoder MLP, triplane.py:113-136, inside).  One launch does, per ray: stratified depth
 * proposals, tri-plane bilinear lookups, the 32->64->33 MLP on the matrix cores, the coarse
 * ray march, importance resampling, the fine pass, the depth merge and the final composite.
 */
typedef struct gnerf_render_params {
    /* planes, NHWC float32 [n_items*3, plane_h, plane_w, 32] (see gnerf_planes_to_nhwc) */
    const float* planes_nhwc;
    int32_t n_items, plane_h, plane_w;
    /* rays: [n_items, rays_per_item, 3] float32 each */
    const float* ray_origins;
    const float* ray_dirs;
    int32_t rays_per_item;
    int32_t image_width;        /* >0: rays of an item form a row-major image of this width (locality hint only) */
    /* decoder: EFFECTIVE weights (FullyConnectedLayer gains folded in, networks_stylegan2.py:118-127),
       row-major w1 [64,32], b1 [64], w2 [33,64], b2 [33]; output 0 is the density */
    const float* w1; const float* b1; const float* w2; const float* b2;
    /* sampling options (rendering_kwargs, renderer.py:91-116) */
    int32_t depth_resolution;             /* coarse samples per ray, 2..GNERF_MAX_SAMPLES */
    int32_t depth_resolution_importance;  /* fine samples per ray, 0..GNERF_MAX_SAMPLES */
    float   ray_start, ray_end;           /* used when ray_start_per_ray is NULL */
    const float* ray_start_per_ray;       /* optional [n_items*rays_per_item] ('auto' box limits, renderer.py:93-98) */
    const float* ray_end_per_ray;
    float   box_warp;
    int32_t white_back;                   /* ray_marcher.py:52-53 */
    int32_t disparity_space_sampling;     /* renderer.py:174-181 */
    /* uniform [0,1) draws, in the reference's order: noise_coarse = rand_like([n,m,S,1]) as [n*m, S],
       noise_fine = rand(n*m, F) (renderer.py:190 and :241).  noise_fine may be NULL iff F == 0 */
    const float* noise_coarse;
    const float* noise_fine;
    /* outputs */
    float* out_rgb;      /* [n_items, rays_per_item, 32]  in (-1,1) */
    float* out_depth;    /* [n_items, rays_per_item, 1] */
    float* out_wsum;     /* [n_items, rays_per_item, 1]   sum of the final weights */
    /* workspace of gnerf_render_workspace_bytes() bytes (holds the call-wide depth range used by the
       global clamp of ray_marcher.py:49-50, or one range per item: depth_clamp_per_item).  ZERO it once after allocation;
       every call leaves it zeroed.
       One workspace per stream that renders concurrently. */
    void*  workspace;
    /* optional stage dump for debugging/parity: float32 [n*m, GNERF_DEBUG_SLOTS, S+F]; NULL in production */
    float* debug;
    /* Decoder arithmetic (the reference's decoder is fp32 addmm, triplane.py:124-136 / networks_stylegan2.py:121-134).
       GNERF_MLP_F32: exact fp32 products on v_mfma_f32_16x16x4_f32, any finite input.
       GNERF_MLP_F16X3: each product as an error-compensated f16 hi/lo split on v_mfma_f32_16x16x32_f16 (fp32-grade, 1.6x
       faster) -- only valid while features, weights and hidden activations are inside f16's range.
       GNERF_MLP_AUTO (0, default): decided ON THE DEVICE per call from max |planes| and the decoder's weights (bounds
       in DESIGN.md section 2.1): out-of-range or ill-conditioned inputs take the fp32 path, so results never depend on
       f16's range.  (AUTO's f16 body also evaluates softplus as log2(1 + 2^p') when the same norms keep every pre-activation
       below exp2's overflow, and in the form that is safe for any p' otherwise -- as a forced GNERF_MLP_F16X3 always does: the
       two forms differ by fp32 rounding.)  planes_absmax: one device float from gnerf_planes_to_nhwc_stats / gnerf_planes_absmax /
       gnerf_upsample2x_add_nhwc; NULL makes the AUTO launcher measure it itself (one extra pass over the planes).
       CONTRACT of a caller-supplied planes_absmax: at the time the render kernel runs (stream order) the float must be
       >= max |planes_nhwc| of THIS call's planes -- an upper bound is fine (it only sends more calls to the fp32 body, or to
       the f16 body's overflow-safe softplus form: the same values up to fp32 rounding), a
       value that is too small (a stale measurement of other or since-modified planes) is NOT detected: the f16x3 body then
       runs on features outside the range its bounds were checked for and can return inf / NaN or lose its fp32-grade
       accuracy.  NaN or +inf there selects the fp32 body.  Debug aid: with the environment variable GNERF_VERIFY_ABSMAX=1
       gnerf_render_forward measures max |planes| itself, synchronises the stream, and fails with GNERF_E_ARG when the
       supplied value is smaller (a test-suite switch: it costs a pass over the planes and a host round trip per call).
       The kernels that always compute in fp32 (more than 144+144 samples or no importance pass beyond 96 samples: the generic
       kernel, gnerf_query_points, both backward passes) ignore these two fields. */
    const float* planes_absmax;
    int32_t mlp_mode;
    /* Layout of `planes_nhwc` (and of gnerf_render_grads.grad_planes_nhwc, which always mirrors it):
       0: [n_items*3, plane_h, plane_w, 32]  one NHWC image per plane (what gnerf_planes_to_nhwc makes);
       1: [n_items, plane_h, plane_w, 96]    the three planes of an item interleaved per texel = the channels_last memory of the
          backbone's own [n_items, 96, plane_h, plane_w] output (triplane.py:69-74), plane p = channels 32p..32p+31.  A
          producer that writes channels_last (gnerf_upsample2x_add_nhwc) feeds the renderer with no layout change at all. */
    int32_t planes_interleaved;
    /* (ABI 6) Batching several views of ONE scene in a call -- gen_videos.py's orbit renders every frame from the same planes:
       planes_shared = 1: `planes_nhwc` holds ONE item's planes ([3,h,w,32] or [1,h,w,96]) and every item of the call reads them
         (n_items still counts the items = views of the call: rays, noise and outputs are per item).  Forward only.
       depth_clamp_per_item = 1: the depth clamp of ray_marcher.py:49-50 (min / max sample depth of the whole forward call) is taken
         per ITEM instead of over the call, so that an item's result does not depend on what it is batched with: a batch of k views
         equals k calls of one view bit for bit.  0 (default): the reference's call-wide clamp.  n_items <= 4096 when set. */
    int32_t planes_shared;
    int32_t depth_clamp_per_item;
    /* (ABI 8) Rays and uniform draws made INSIDE the render kernel (SURVEY.md section 8a row 1 / 8d "in-kernel Philox"): the call then
       reads no ray tensors and no noise tensors (26.7 MB per 65 536 rays at 48+48) and needs no launch in front of it.
       cam2world / intrinsics: [n_items,4,4] and [n_items,3,3] row-major device floats, used when ray_origins == ray_dirs == NULL.
         Ray m of an item looks through pixel centre ((m % w + .5) / w, (m / w + .5) / w), w = image_width, which must be > 0 with
         rays_per_item == w * w: the arithmetic of gnerf_make_rays (ray_sampler.py:24-63), same bits.
       rng_mode = GNERF_RNG_TORCH_PHILOX: noise_coarse / noise_fine must be NULL; the kernel draws what torch's device generator
         would have put into them -- `torch.rand_like([n,m,S,1])` then `torch.rand(n*m, F)` (renderer.py:190,241) -- Philox4x32-10
         keyed by rng_seed, element li of a draw taken from thread li % threads of ATen's grid-stride kernel at the generator's
         philox offset (oracle/philox_ref.py states the recipe; gnerf_torch_rand_plan gives threads and the offset increment of a
         draw).  rng_offset_coarse / rng_offset_fine: philox offset of the generator BEFORE the respective draw (multiples of 4).
         The caller advances its generator by the two increments afterwards, exactly as the two torch.rand calls would have.
       rng_per_item = 1: the draws are per ITEM instead of per call (what n_items separate calls with one item each would draw:
         the batched views of planes_shared): element indices count inside the item, and item i's offsets are
         rng_offset_* + i * rng_offset_item_stride.
       Supported by the pipelined kernels at their compile-time sample counts (48+48 and 96+96, no disparity sampling, no per-ray
       limits, no stage dump): anything else returns GNERF_E_UNSUPPORTED and the caller passes tensors.  Forward only. */
    const float* cam2world;
    const float* intrinsics;
    int32_t  rng_mode;
    int32_t  rng_per_item;
    uint64_t rng_seed;
    uint64_t rng_offset_coarse, rng_offset_fine, rng_offset_item_stride;
    uint32_t rng_threads_coarse, rng_threads_fine;
    /* (ABI 10) density noise (renderer.py:146-147: `sigma += randn_like(sigma) * density_noise` in run_model, i.e. once on the coarse
       pass's densities and once on the fine pass's, before either ray march): the two tensors ALREADY multiplied by density_noise,
       float32 [n_items * rays_per_item, depth_resolution] and [.., depth_resolution_importance], or NULL (off).  Forward only
       (gnerf_render_backward returns GNERF_E_UNSUPPORTED); not with in-kernel rays / draws. */
    const float* sigma_noise_coarse;
    const float* sigma_noise_fine;
} gnerf_render_params;

#define GNERF_RNG_TENSORS      0
#define GNERF_RNG_TORCH_PHILOX 1

/* Launch geometry of ATen's uniform kernel for a draw of `numel` floats on a device whose torch properties are
 * multi_processor_count / max_threads_per_multi_processor: *threads = 256 * min(ceil(numel / 256), mpc * (mtpm / 256)) and
 * *offset_increment = 4 * ceil(numel / (4 * threads)), what the generator's philox offset advances by.  Host-only arithmetic. */
int gnerf_torch_rand_plan(int64_t numel, int multi_processor_count, int max_threads_per_multi_processor,
                          uint32_t* threads, uint64_t* offset_increment);
/* The draw itself as a stand-alone kernel: out[i] = element i of torch.rand(numel) at (seed, offset) -- the same device function the
 * render kernels use, exposed so that it can be held to torch.rand directly (tests) and for callers that want tensors. */
int gnerf_torch_rand(float* out, int64_t numel, uint64_t seed, uint64_t offset, uint32_t threads, gnerf_stream_t stream);
/* (ABI 10) The rays of gnerf_make_rays AND the renderer's two uniform draws in one launch: draw_a / draw_b = what torch.rand(numel_a) then
 * torch.rand(numel_b) return with the device generator at seed and philox offsets offset_a / offset_b (threads_* from gnerf_torch_rand_plan;
 * the caller advances its generator by the two increments, as for gnerf_render_params.rng_mode).  draw_b may be NULL.  One Philox block per
 * four elements, ATen's own thread-to-element map: the same values as gnerf_torch_rand, i.e. as torch.rand, bit for bit. */
int gnerf_make_rays_and_draws(const float* cam2world, const float* intrinsics, int n, int res, float* origins, float* dirs,
                              float* draw_a, int64_t numel_a, uint64_t offset_a, uint32_t threads_a,
                              float* draw_b, int64_t numel_b, uint64_t offset_b, uint32_t threads_b, uint64_t seed, gnerf_stream_t stream);

#define GNERF_MLP_AUTO  0
#define GNERF_MLP_F16X3 1
#define GNERF_MLP_F32   2

#define GNERF_MAX_SAMPLES   256
#define GNERF_DEBUG_SLOTS   8
/* debug slots */
#define GNERF_DBG_DEPTH_COARSE 0
#define GNERF_DBG_SIGMA_COARSE 1
#define GNERF_DBG_WEIGHT_COARSE 2
#define GNERF_DBG_DEPTH_FINE   3
#define GNERF_DBG_SIGMA_FINE   4
#define GNERF_DBG_DEPTH_SORTED 5
#define GNERF_DBG_SIGMA_SORTED 6
#define GNERF_DBG_WEIGHT_FINAL 7

size_t gnerf_render_workspace_bytes(void);
int    gnerf_render_forward(const gnerf_render_params* p, gnerf_stream_t stream);

/* Gradient of the renderer: what autograd computes upstream by walking the graph of
 * renderer.py:88-140 backwards (grid_sample_gradfix.py's backward op for the planes, the decoder's
 * linear layers, ray_marcher.py's composite).  The forward pass is recomputed per ray from the same
 * params (same noise), nothing is saved between the two calls.  Importance depths are constants, as
 * upstream (renderer.py:198 no_grad / :211 detach); rays and noise get no gradient.
 * All output buffers are ACCUMULATED into (the caller zeroes them, or keeps summing across calls). */
typedef struct gnerf_render_grads {
    const float* grad_rgb;      /* [n_items, rays_per_item, 32] or NULL (= zeros) */
    const float* grad_depth;    /* [n_items, rays_per_item, 1]  or NULL */
    const float* grad_wsum;     /* [n_items, rays_per_item, 1]  or NULL */
    float* grad_planes_nhwc;    /* [n_items*3, plane_h, plane_w, 32] or NULL (skip) */
    float* grad_w1;             /* [64,32]; the four decoder gradients are given together or all NULL */
    float* grad_b1;             /* [64] */
    float* grad_w2;             /* [33,64] */
    float* grad_b2;             /* [33] */
    /* Optional workspace (contents irrelevant before and after the call; 256-byte aligned).  The struct carries no size field: what the
       buffer must hold follows from the request, and the call trusts that it was sized by the matching function for the SAME params:
         * with grad_planes_nhwc: gnerf_render_backward_stage_bytes(p) bytes; the plane gradient is made in two passes.  Layout: [rays * (S + F)] rows of 33
           floats (32 feature gradients in the ray's depth order + one spare), one spare 256-byte line, then (ABI 9) the binned
           scatter's workspace: per-plane-tile counts / starts / scales / order, the record arrays (8 + 16 bytes per (sample, plane)
           record) and the tile halos.  The second pass sums every 16 x 16-texel plane tile in LDS in 64-bit fixed point and writes it
           with plain stores -- no float atomics, bit-identical gradients from run to run (GNERF_BWD_SCATTER=sorted selects round 4's
           LDS-merged atomic pass instead, which uses only the rows);
         * without grad_planes_nhwc (decoder gradients only): gnerf_render_backward_exchange_bytes(p) bytes suffice (below): per sample
           depth, colour weight, dL/dsigma.
       NULL: single pass (one float atomic per tap and channel). */
    float* scatter_stage;
} gnerf_render_grads;

size_t gnerf_render_backward_stage_bytes(const gnerf_render_params* p);
/* (ABI 8) A decoder-only request (grad_planes_nhwc NULL) may pass a much smaller buffer of this many bytes as scatter_stage: what the
 * backward's two kernels on the pipelined path exchange per sample (depth, colour weight, dL/dsigma).  Without it such a request
 * runs the one-wave-per-ray kernel. */
size_t gnerf_render_backward_exchange_bytes(const gnerf_render_params* p);

/* p: the forward call's params (outputs, workspace and debug are ignored). */
int gnerf_render_backward(const gnerf_render_params* p, const gnerf_render_grads* g, gnerf_stream_t stream);

/* Density / colour of arbitrary points (run_model, renderer.py:142-148; used by
 * TriPlaneGenerator.sample / sample_mixed for shape extraction):
 * points [n_items, n_points, 3] -> sigma [n_items, n_points, 1], rgb [n_items, n_points, 32].
 * out_rgb may be NULL: densities only (the 512^3 shape extraction of gen_videos.py:189-224 reads nothing else).
 * planes_interleaved: layout of planes_nhwc, as gnerf_render_params.planes_interleaved. */
int gnerf_query_points(const float* planes_nhwc, int n_items, int plane_h, int plane_w,
                       const float* points, int n_points, float box_warp,
                       const float* w1, const float* b1, const float* w2, const float* b2,
                       float* out_sigma, float* out_rgb, int planes_interleaved, gnerf_stream_t stream);

/* Gradient of gnerf_query_points: grad_sigma [n_items, n_points, 1] and grad_rgb [n_items, n_points, 32] (either may be
 * NULL) -> ACCUMULATED into grad_planes_nhwc and the four decoder gradients (same conventions as gnerf_render_backward).
 * Points get no gradient.  Upstream: autograd through renderer.py:142-148 (sample_mixed in the density regulariser). */
int gnerf_query_points_backward(const float* planes_nhwc, int n_items, int plane_h, int plane_w,
                                const float* points, int n_points, float box_warp,
                                const float* w1, const float* b1, const float* w2, const float* b2,
                                const float* grad_sigma, const float* grad_rgb,
                                float* grad_planes_nhwc, float* grad_w1, float* grad_b1, float* grad_w2, float* grad_b2,
                                int planes_interleaved, gnerf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GNERF_HIP_H */
