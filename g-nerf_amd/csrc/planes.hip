// Layout change and ray generation feeding the fused renderer.
//
// gnerf_planes_to_nhwc: the reference keeps tri-planes NCHW (training/triplane.py:74), where the 32 channels
// of one texel are 256 KB apart.  The renderer wants one texel = one 128-byte line, so planes are
// transposed once per batch to [plane, y, x, channel] through a 32x64 LDS tile: reads are coalesced
// along x, writes are whole texels.
//
// gnerf_make_rays: RaySampler.forward (training/volumetric_rendering/ray_sampler.py:24-63).

#include "common.h"
#include "raygen.h"

namespace {

using namespace gnerf;

// (Round 6, profiles/r06_repack_tile_ab.jsonl: 32 x 128 and 32 x 256 tiles -- 16 / 32 dword loads in flight per lane instead of 8 -- move the
//  headline step by -0.4 %: the pass is not short of bytes in flight; at 200 MB in 50 us it runs at the rate HBM gives a read + write stream --
//  profiles/r06_step_copy_bound.json: in the step it costs 51.8 us incl. its memset where torch's copy kernel on the same 100 MB costs 45.3.)
// GNERF_REPACK_NT (experiment, profiles/r06_repack_nt_ab.jsonl): 1 = the NCHW source read non-temporal, 2 = the result stored so, 3 = both: the
// headline step gets SLOWER by 15 / 14 / 31 us of 553 -- at the default policy a part of the source still comes from the Infinity Cache and the
// render kernel finds the result there; off
#ifndef GNERF_REPACK_NT
#define GNERF_REPACK_NT 0
#endif
#ifndef GNERF_REPACK_PX
#define GNERF_REPACK_PX 64
#endif
constexpr int CT = 32, PT = GNERF_REPACK_PX;   // channel x pixel tile (PT / 8 dword loads in flight per lane)

// max |x| as an unsigned compare of the sign-stripped bits: orders like the floats for finite values and +inf, and any
// NaN compares above +inf, so a NaN in the input survives as a NaN in the result.
__device__ __forceinline__ unsigned abs_bits(float v) { return __float_as_uint(v) & 0x7fffffffu; }
// One atomic per WORKGROUP at most, and only when it would raise the published value: thousands of same-address atomics
// serialise at the memory side (49 152 of them -- one per wave -- made the 37 us repack take 0.5 ms).  The pre-check reads the
// word with a device-scope atomic load; a stale or racing read only costs a redundant atomicMax, never a wrong result.
__device__ __forceinline__ void publish_absmax(unsigned m, unsigned* out) {
    __shared__ unsigned wave_max[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned b = max(max(wave_max[0], wave_max[1]), max(wave_max[2], wave_max[3]));
        if (b > __hip_atomic_load(out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(out, b);
    }
}

// (Measured and dropped, round 3: the same tile with 16-byte global accesses on both sides -- a lane reads four pixels of a channel and
// writes four channels of a pixel.  Alone it is faster, 34.5 -> 29.8 us for 2 x 100 MB on a warm Infinity Cache; inside bench.py's step,
// where the source comes from HBM and the render kernel reads the result next, the step got SLOWER: 0.578 -> 0.591 ms, two runs each
// way on one box, render call unchanged.  The dword form stays.)
template <bool STATS>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                           int c, int64_t hw, int tiles_p, int tiles_c, unsigned* absmax) {
    __shared__ float tile[CT][PT + 1];
    int64_t t = blockIdx.x;
    const int tp = int(t % tiles_p); t /= tiles_p;
    const int tc = int(t % tiles_c); t /= tiles_c;
    const int64_t plane = t;
    const int64_t p0 = int64_t(tp) * PT;
    const int c0 = tc * CT;
    const float* s = src + plane * c * hw;
    float* d = dst + plane * hw * c;
    unsigned amax = 0u;
#pragma unroll
    for (int i = 0; i < CT * PT / 256; i++) {
        const int e = threadIdx.x + 256 * i;
        const int ch = e / PT, px = e % PT;
        float v = 0.f;
        if (c0 + ch < c && p0 + px < hw) v = (GNERF_REPACK_NT & 1) ? __builtin_nontemporal_load(s + int64_t(c0 + ch) * hw + p0 + px) : s[int64_t(c0 + ch) * hw + p0 + px];
        tile[ch][px] = v;
        if (STATS) amax = max(amax, abs_bits(v));
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < CT * PT / 256; i++) {
        const int e = threadIdx.x + 256 * i;
        const int px = e / CT, ch = e % CT;
        if (c0 + ch < c && p0 + px < hw) { if (GNERF_REPACK_NT & 2) __builtin_nontemporal_store(tile[ch][px], d + (p0 + px) * c + c0 + ch); else d[(p0 + px) * c + c0 + ch] = tile[ch][px]; }
    }
    if (STATS) publish_absmax(amax, absmax);
}

// Eight 16-byte loads in flight per lane on every trip, the ragged end included: an index past the end is clamped onto the last
// vector (a duplicate does not change a maximum), so there is no dependent tail loop -- the first version's tail (two or three
// loads issued one after the other by every lane of a 100 MB reduction) held it at 2.4 TB/s.
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, int64_t numel, unsigned* absmax) {
    unsigned amax = 0u;
    const int64_t n4 = numel >> 2;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    const int64_t stride = int64_t(gridDim.x) * 256;
    if (n4 > 0) {
        for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < n4; i += 8 * stride) {
            float4 v[8];
#pragma unroll
            for (int k = 0; k < 8; k++) { const int64_t idx = i + k * stride; v[k] = x4[idx < n4 ? idx : n4 - 1]; }
#pragma unroll
            for (int k = 0; k < 8; k++) amax = max(max(amax, abs_bits(v[k].x)), max(max(abs_bits(v[k].y), abs_bits(v[k].z)), abs_bits(v[k].w)));
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < (numel & 3)) amax = max(amax, abs_bits(x[(n4 << 2) + threadIdx.x]));
    publish_absmax(amax, absmax);
}

// ---------------------------------------------------------------------------------------------
// The tri-plane PRODUCER's last step, fused and writing the renderer's layout directly.
// The backbone's final block ends with  img = upsample2d(img) + torgb(x)  (networks_stylegan2.py:456-463: the 4x4 low-pass
// filter after x2 zero insertion, gain 4, then an add), and triplane.py:74 views the result as three 32-channel planes.
// This kernel computes that sum and stores it CHANNELS-LAST: [n, 2h, 2w, c] memory = the interleaved plane layout the render
// kernels address directly (gnerf_render_params.planes_interleaved), plus max |planes| for the decoder-arithmetic choice --
// in place of an NCHW upsample, an NCHW add, and a 100 MB layout change.
// Workgroup = 4 x 32 output pixels x 32 channels.  Phase A (lanes along x: coalesced NCHW reads of y, input window of img in
// LDS) makes the sums into an LDS tile; phase B (lanes along channels) writes whole 128-byte channel groups.
struct Up2Taps { float k[4][4]; };      // effective taps: gain * (f flipped unless flip)

constexpr int kUpCh = 32, kUpRows = 4, kUpCols = 32, kUpInPitch = 4 * 18 + 1, kUpOutPitch = kUpCh + 4;      // (output rows 16-byte aligned: vector hand-off)

__global__ __launch_bounds__(256) void upsample2x_add_nhwc_kernel(const float* __restrict__ img, const float* __restrict__ y, Up2Taps K,
                                                                  float* __restrict__ out, int c, int h, int w, unsigned* absmax) {
    __shared__ float in[kUpCh * kUpInPitch];
    __shared__ __align__(16) float ot[kUpRows * kUpCols * kUpOutPitch];
    const int OH = 2 * h, OW = 2 * w;
    const int tiles_x = OW / kUpCols, tiles_y = OH / kUpRows, groups = c / kUpCh;
    // XCD-contiguous numbering (workgroups go to the eight XCDs round-robin, each with its own L2): vertical neighbours share two of
    // their four input rows of `img`, horizontal ones two columns
    int b = blockIdx.x;
    if (gridDim.x % kNumXCD == 0) b = (blockIdx.x % kNumXCD) * (gridDim.x / kNumXCD) + blockIdx.x / kNumXCD;
    const int tx0 = b % tiles_x; b /= tiles_x;
    const int ty0 = b % tiles_y; b /= tiles_y;
    const int cg0 = b % groups;  b /= groups;
    const int n = b;
    const int ox0 = tx0 * kUpCols, oy0 = ty0 * kUpRows, c0 = cg0 * kUpCh;
    const int ix0 = ox0 / 2 - 1, iy0 = oy0 / 2 - 1;
    const float* img_n = img + (int64_t(n) * c + c0) * h * w;
    for (int e = threadIdx.x; e < kUpCh * 4 * 18; e += 256) {
        const int ch = e / 72, r = (e % 72) / 18, cc = e % 18;
        const int iy = iy0 + r, ix = ix0 + cc;
        float v = 0.f;
        if (iy >= 0 && iy < h && ix >= 0 && ix < w) v = img_n[(int64_t(ch) * h + iy) * w + ix];
        in[ch * kUpInPitch + r * 18 + cc] = v;
    }
    __syncthreads();
    // Phase A (round 4: 16-byte global accesses in both phases; 3.4 -> see profiles/r04_ops_GBs.jsonl): a lane makes four consecutive
    // output columns of four channels -- y arrives as one float4 per channel -- and hands them over as one float4 of channels per pixel.
    {
        const int x4 = threadIdx.x & 7, ty = (threadIdx.x >> 3) & 3, chq = threadIdx.x >> 5;
        const int ry = ty & 1, r = (ty >> 1) + ry;
        const int oy = oy0 + ty;
        float v[4][4];                                   // [channel][column]
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int ch = chq * 4 + q;
            const float* win = in + ch * kUpInPitch + r * 18 + 2 * x4;
            const float a0 = win[0], a1 = win[1], a2 = win[2], a3 = win[3], b0 = win[18], b1 = win[19], b2 = win[20], b3 = win[21];
            // column i of the four: rx = i & 1, window column (i >> 1) + rx; taps K[ry][rx], K[ry][rx + 2], K[ry + 2][rx], K[ry + 2][rx + 2]
            v[q][0] = K.k[ry][0] * a0 + K.k[ry][2] * a1 + K.k[ry + 2][0] * b0 + K.k[ry + 2][2] * b1;
            v[q][1] = K.k[ry][1] * a1 + K.k[ry][3] * a2 + K.k[ry + 2][1] * b1 + K.k[ry + 2][3] * b2;
            v[q][2] = K.k[ry][0] * a1 + K.k[ry][2] * a2 + K.k[ry + 2][0] * b1 + K.k[ry + 2][2] * b2;
            v[q][3] = K.k[ry][1] * a2 + K.k[ry][3] * a3 + K.k[ry + 2][1] * b2 + K.k[ry + 2][3] * b3;
            if (y) {
                const float4 yy = *reinterpret_cast<const float4*>(y + ((int64_t(n) * c + c0 + ch) * OH + oy) * OW + ox0 + 4 * x4);
                v[q][0] += yy.x; v[q][1] += yy.y; v[q][2] += yy.z; v[q][3] += yy.w;
            }
        }
#pragma unroll
        for (int i = 0; i < 4; i++)
            *reinterpret_cast<float4*>(ot + (ty * kUpCols + 4 * x4 + i) * kUpOutPitch + 4 * chq) = make_float4(v[0][i], v[1][i], v[2][i], v[3][i]);
    }
    __syncthreads();
    // Phase B: eight lanes write one pixel's 32 channels (128 contiguous bytes), a float4 each
    unsigned amax = 0u;
    {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int item = threadIdx.x + 256 * q;
            const int ch4 = item & 7, px = item >> 3;            // pixel of the tile: row px / 32, column px % 32
            const float4 vv = *reinterpret_cast<const float4*>(ot + px * kUpOutPitch + 4 * ch4);
            *reinterpret_cast<float4*>(out + ((int64_t(n) * OH + oy0 + (px >> 5)) * OW + ox0 + (px & 31)) * c + c0 + 4 * ch4) = vv;
            amax = max(max(amax, abs_bits(vv.x)), max(max(abs_bits(vv.y), abs_bits(vv.z)), abs_bits(vv.w)));
        }
    }
    if (absmax) publish_absmax(amax, absmax);
}

// The inverse layout change, for the gradient of the planes: [plane, y, x, channel] -> [plane, channel, y, x].
// Reads whole texels, writes rows along x.
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                           int c, int64_t hw, int tiles_p, int tiles_c) {
    __shared__ float tile[PT][CT + 1];
    int64_t t = blockIdx.x;
    const int tp = int(t % tiles_p); t /= tiles_p;
    const int tc = int(t % tiles_c); t /= tiles_c;
    const int64_t plane = t;
    const int64_t p0 = int64_t(tp) * PT;
    const int c0 = tc * CT;
    const float* s = src + plane * hw * c;
    float* d = dst + plane * c * hw;
    for (int e = threadIdx.x; e < CT * PT; e += 256) {
        const int px = e / CT, ch = e % CT;
        float v = 0.f;
        if (c0 + ch < c && p0 + px < hw) v = s[(p0 + px) * c + c0 + ch];
        tile[px][ch] = v;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < CT * PT; e += 256) {
        const int ch = e / PT, px = e % PT;
        if (c0 + ch < c && p0 + px < hw) d[int64_t(c0 + ch) * hw + p0 + px] = tile[px][ch];
    }
}

// One lane per ray; the arithmetic is camera_ray's (raygen.h), shared with the render kernels' in-kernel form.
__global__ __launch_bounds__(256) void make_rays_kernel(const float* __restrict__ c2w, const float* __restrict__ intr,
                                                        int n, int res, float* __restrict__ origins, float* __restrict__ dirs) {
    const int64_t m_total = int64_t(res) * res;
    const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (i >= m_total * n) return;
    const int item = int(i / m_total);
    const int m = int(i % m_total);
    const float* M = c2w + item * 16;
    float d[3];
    camera_ray(M, intr + item * 9, res, m / res, m % res, d);
#pragma unroll
    for (int r = 0; r < 3; r++) {
        dirs[i * 3 + r] = d[r];
        origins[i * 3 + r] = M[r * 4 + 3];
    }
}

// Round 6: what a render step needs in front of the fused kernel -- the rays of n cameras and the reference's two uniform draws (renderer.py:
// 190 `rand_like([N,M,S,1])`, :241 `rand(N*M, F)`) -- in ONE launch instead of three (gnerf_make_rays + two torch.rand).  The draws are
// torch.rand's, bit for bit: thread t of a draw evaluates the Philox block of (counter ctr + call, subsequence t) and stores its four words
// at elements t + threads * (4 call + word), exactly as ATen's grid-stride kernel does (raygen.h: torch_rand_element is the per-element view of
// the same map) -- one block per four elements, where the stand-alone gnerf_torch_rand evaluates one per element.
__global__ __launch_bounds__(256) void rays_and_draws_kernel(const float* __restrict__ c2w, const float* __restrict__ intr, int n, int res,
                                                             float* __restrict__ origins, float* __restrict__ dirs,
                                                             float* __restrict__ out_a, int64_t numel_a, TorchRandDraw da, uint32_t threads_a,
                                                             float* __restrict__ out_b, int64_t numel_b, TorchRandDraw db, uint32_t threads_b) {
    const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
    const int64_t m_total = int64_t(res) * res;
    if (i < m_total * n) {
        const int item = int(i / m_total);
        const int m = int(i % m_total);
        const float* M = c2w + item * 16;
        float d[3];
        camera_ray(M, intr + item * 9, res, m / res, m % res, d);
#pragma unroll
        for (int r = 0; r < 3; r++) {
            dirs[i * 3 + r] = d[r];
            origins[i * 3 + r] = M[r * 4 + 3];
        }
    }
    auto draw = [&](float* __restrict__ out, int64_t numel, const TorchRandDraw& d, uint32_t threads) {
        if (!out || i >= int64_t(threads)) return;
        const int64_t per_thread = (numel - i + int64_t(threads) - 1) / int64_t(threads);      // elements of this thread: i, i + threads, ...
        for (int64_t call = 0; 4 * call < per_thread; call++) {
            const uint64_t c = d.ctr + uint64_t(call);
            uint32_t w[4];
            philox4x32_10_block(uint32_t(c), uint32_t(c >> 32), uint32_t(i), 0u, d.k0, d.k1, w);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int64_t li = i + int64_t(threads) * (4 * call + k);
                if (li < numel) {
                    const float u = __fmaf_rn(float(w[k]), 0x1p-32f, 0x1p-32f);
                    out[li] = u == 1.0f ? 0.0f : u;
                }
            }
        }
    };
    draw(out_a, numel_a, da, threads_a);
    draw(out_b, numel_b, db, threads_b);
}

// torch.rand(numel) at (seed, offset) as a stand-alone kernel: one element per lane, grid-stride free (numel <= 2^32 checked by the host)
__global__ __launch_bounds__(256) void torch_rand_kernel(float* __restrict__ out, int64_t numel, TorchRandDraw d) {
    const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (i < numel) out[i] = torch_rand_element(d, uint64_t(i));
}

// gen_videos.py:173 -- `(img * 127.5 + 128).clamp(0, 255).to(torch.uint8)` followed by the NCHW -> NHWC permute of the frame writer, in one
// pass: one thread per pixel reads its c channel values (coalesced along x per channel plane) and writes c adjacent bytes.  The product
// and the sum are rounded separately, as the two PyTorch ops round them; the cast truncates; NaN becomes 0.
__global__ __launch_bounds__(256) void to_uint8_nhwc_kernel(const float* __restrict__ img, unsigned char* __restrict__ out, int c, int64_t hw, int64_t total) {
    const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;          // (image, pixel)
    if (i >= total) return;
    const int64_t n = i / hw, p = i - n * hw;
    const float* src = img + n * c * hw + p;
    unsigned char* dst = out + i * c;
    for (int ch = 0; ch < c; ch++) {
        float m = src[int64_t(ch) * hw] * 127.5f;
        asm volatile("" : "+v"(m));                                 // the product is rounded on its own, as torch's mul kernel rounds it: no FMA with the add
        const float v = m + 128.f;
        dst[ch] = (unsigned char)fminf(fmaxf(v, 0.f), 255.f);
    }
}

}  // namespace

extern "C" int gnerf_to_uint8_nhwc(const float* img, unsigned char* out, int n, int c, int h, int w, gnerf_stream_t stream) {
    using namespace gnerf;
    if (!img || !out) return fail(GNERF_E_ARG, "to_uint8_nhwc: null pointer");
    if (n < 1 || c < 1 || c > 64 || h < 1 || w < 1) return fail(GNERF_E_ARG, "to_uint8_nhwc: bad shape (1 <= channels <= 64)");
    const int64_t hw = int64_t(h) * w, total = hw * n;
    const int64_t blocks = (total + 255) / 256;
    if (blocks > INT32_MAX) return fail(GNERF_E_ARG, "to_uint8_nhwc: tensor too large");
    hipLaunchKernelGGL(to_uint8_nhwc_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), img, out, c, hw, total);
    return check_launch("to_uint8_nhwc");
}

static int planes_to_nhwc_impl(const float* planes_nchw, float* planes_nhwc, int np, int c, int h, int w, float* absmax, bool stats,
                               gnerf_stream_t stream) {
    using namespace gnerf;
    if (!planes_nchw || !planes_nhwc || (stats && !absmax)) return fail(GNERF_E_ARG, "planes_to_nhwc: null pointer");
    if (np < 1 || c < 1 || h < 1 || w < 1) return fail(GNERF_E_ARG, "planes_to_nhwc: empty tensor");
    const int64_t hw = int64_t(h) * w;
    const int tiles_p = int((hw + PT - 1) / PT), tiles_c = (c + CT - 1) / CT;
    const int64_t blocks = int64_t(tiles_p) * tiles_c * np;
    if (blocks > INT32_MAX) return fail(GNERF_E_ARG, "planes_to_nhwc: tensor too large");
    if (stats) {
        if (hipMemsetAsync(absmax, 0, sizeof(float), as_stream(stream)) != hipSuccess) return fail(GNERF_E_LAUNCH, "planes_to_nhwc: memset failed");
        hipLaunchKernelGGL(nchw_to_nhwc_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream),
                           planes_nchw, planes_nhwc, c, hw, tiles_p, tiles_c, reinterpret_cast<unsigned*>(absmax));
    } else {
        hipLaunchKernelGGL(nchw_to_nhwc_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream),
                           planes_nchw, planes_nhwc, c, hw, tiles_p, tiles_c, static_cast<unsigned*>(nullptr));
    }
    return check_launch("planes_to_nhwc");
}

extern "C" int gnerf_planes_to_nhwc(const float* planes_nchw, float* planes_nhwc, int np, int c, int h, int w,
                                    gnerf_stream_t stream) {
    return planes_to_nhwc_impl(planes_nchw, planes_nhwc, np, c, h, w, nullptr, false, stream);
}

extern "C" int gnerf_planes_to_nhwc_stats(const float* planes_nchw, float* planes_nhwc, int np, int c, int h, int w,
                                          float* absmax, gnerf_stream_t stream) {
    return planes_to_nhwc_impl(planes_nchw, planes_nhwc, np, c, h, w, absmax, true, stream);
}

extern "C" int gnerf_upsample2x_add_nhwc(const float* img, const float* y, const float* f_host, int flip, float gain, float* out,
                                         int n, int c, int h, int w, float* absmax, gnerf_stream_t stream) {
    using namespace gnerf;
    if (!img || !f_host || !out) return fail(GNERF_E_ARG, "upsample2x_add_nhwc: null pointer");
    if (n < 1 || c < 1 || h < 1 || w < 1) return fail(GNERF_E_ARG, "upsample2x_add_nhwc: empty tensor");
    if (c % kUpCh != 0 || (2 * w) % kUpCols != 0 || (2 * h) % kUpRows != 0) return fail(GNERF_E_UNSUPPORTED, "upsample2x_add_nhwc: needs channels %% 32 == 0, width %% 16 == 0, height %% 2 == 0");
    // the kernel reads y and writes out as 16-byte vectors: a contiguous view at a storage offset that is no multiple of four floats
    // is not for it (the wrapper composes the PyTorch ops then, as for any other unsupported shape)
    if ((y && (reinterpret_cast<uintptr_t>(y) & 15)) || (reinterpret_cast<uintptr_t>(out) & 15))
        return fail(GNERF_E_UNSUPPORTED, "upsample2x_add_nhwc: y and out must be 16-byte aligned");
    const int64_t blocks = int64_t(n) * (c / kUpCh) * (2 * h / kUpRows) * (2 * w / kUpCols);
    if (blocks > INT32_MAX || int64_t(n) * c * 4 * h * w > INT32_MAX * int64_t(2)) return fail(GNERF_E_ARG, "upsample2x_add_nhwc: tensor too large");
    Up2Taps K;
    for (int ky = 0; ky < 4; ky++)
        for (int kx = 0; kx < 4; kx++) K.k[ky][kx] = gain * (flip ? f_host[ky * 4 + kx] : f_host[(3 - ky) * 4 + (3 - kx)]);
    if (absmax && hipMemsetAsync(absmax, 0, sizeof(float), as_stream(stream)) != hipSuccess) return fail(GNERF_E_LAUNCH, "upsample2x_add_nhwc: memset failed");
    hipLaunchKernelGGL(upsample2x_add_nhwc_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), img, y, K, out, c, h, w,
                       reinterpret_cast<unsigned*>(absmax));
    return check_launch("upsample2x_add_nhwc");
}

extern "C" int gnerf_planes_absmax(const float* planes, int64_t numel, float* absmax, gnerf_stream_t stream) {
    using namespace gnerf;
    if (!planes || !absmax) return fail(GNERF_E_ARG, "planes_absmax: null pointer");
    if (numel < 1) return fail(GNERF_E_ARG, "planes_absmax: empty tensor");
    if (reinterpret_cast<uintptr_t>(planes) & 15) return fail(GNERF_E_ARG, "planes_absmax: planes must be 16-byte aligned");
    if (hipMemsetAsync(absmax, 0, sizeof(float), as_stream(stream)) != hipSuccess) return fail(GNERF_E_LAUNCH, "planes_absmax: memset failed");
    // every lane makes `trips` trips of eight vectors: size the grid so that the trips cover the tensor without a mostly-clamped last one
    const int64_t n4 = numel / 4, per_trip = int64_t(256) * 8;
    const int64_t trips = (n4 + per_trip * kNumCU * 8 - 1) / (per_trip * kNumCU * 8);
    int64_t blocks = trips > 0 ? (n4 + per_trip * trips - 1) / (per_trip * trips) : 1;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), planes, numel, reinterpret_cast<unsigned*>(absmax));
    return check_launch("planes_absmax");
}

extern "C" int gnerf_planes_from_nhwc(const float* planes_nhwc, float* planes_nchw, int np, int c, int h, int w,
                                      gnerf_stream_t stream) {
    using namespace gnerf;
    if (!planes_nchw || !planes_nhwc) return fail(GNERF_E_ARG, "planes_from_nhwc: null pointer");
    if (np < 1 || c < 1 || h < 1 || w < 1) return fail(GNERF_E_ARG, "planes_from_nhwc: empty tensor");
    const int64_t hw = int64_t(h) * w;
    const int tiles_p = int((hw + PT - 1) / PT), tiles_c = (c + CT - 1) / CT;
    const int64_t blocks = int64_t(tiles_p) * tiles_c * np;
    if (blocks > INT32_MAX) return fail(GNERF_E_ARG, "planes_from_nhwc: tensor too large");
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream),
                       planes_nhwc, planes_nchw, c, hw, tiles_p, tiles_c);
    return check_launch("planes_from_nhwc");
}

extern "C" int gnerf_make_rays(const float* cam2world, const float* intrinsics, int n, int res,
                               float* origins, float* dirs, gnerf_stream_t stream) {
    using namespace gnerf;
    if (!cam2world || !intrinsics || !origins || !dirs) return fail(GNERF_E_ARG, "make_rays: null pointer");
    if (n < 1 || res < 1) return fail(GNERF_E_ARG, "make_rays: n and res must be positive");
    const int64_t total = int64_t(n) * res * res;
    hipLaunchKernelGGL(make_rays_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream),
                       cam2world, intrinsics, n, res, origins, dirs);
    return check_launch("make_rays");
}

extern "C" int gnerf_torch_rand_plan(int64_t numel, int multi_processor_count, int max_threads_per_multi_processor,
                                     uint32_t* threads, uint64_t* offset_increment) {
    using namespace gnerf;
    if (numel < 1 || multi_processor_count < 1 || max_threads_per_multi_processor < 256 || !threads || !offset_increment)
        return fail(GNERF_E_ARG, "torch_rand_plan: bad arguments");
    // ATen/native/cuda/DistributionTemplates.h, calc_execution_policy (block 256, unroll 4)
    const uint64_t blocks_cap = uint64_t(multi_processor_count) * uint64_t(max_threads_per_multi_processor / 256);
    uint64_t blocks = (uint64_t(numel) + 255) / 256;
    if (blocks > blocks_cap) blocks = blocks_cap;
    if (blocks * 256 > 0xffffffffull) return fail(GNERF_E_UNSUPPORTED, "torch_rand_plan: grid too large");
    *threads = uint32_t(blocks * 256);
    *offset_increment = ((uint64_t(numel) - 1) / (uint64_t(*threads) * 4) + 1) * 4;
    return GNERF_OK;
}

extern "C" int gnerf_make_rays_and_draws(const float* cam2world, const float* intrinsics, int n, int res, float* origins, float* dirs,
                                         float* draw_a, int64_t numel_a, uint64_t offset_a, uint32_t threads_a,
                                         float* draw_b, int64_t numel_b, uint64_t offset_b, uint32_t threads_b, uint64_t seed, gnerf_stream_t stream) {
    using namespace gnerf;
    if (!cam2world || !intrinsics || !origins || !dirs) return fail(GNERF_E_ARG, "make_rays_and_draws: null pointer");
    if (n < 1 || res < 1) return fail(GNERF_E_ARG, "make_rays_and_draws: n and res must be positive");
    if (!draw_a || numel_a < 1 || (draw_b && numel_b < 1)) return fail(GNERF_E_ARG, "make_rays_and_draws: the first draw must be given, and a second one must not be empty");
    TorchRandDraw da, db = TorchRandDraw{};
    if (!torch_rand_draw(seed, offset_a, threads_a, numel_a, da) || (draw_b && !torch_rand_draw(seed, offset_b, threads_b, numel_b, db)))
        return fail(GNERF_E_UNSUPPORTED, "make_rays_and_draws: the generator geometry (threads %u / %u at offsets %llu / %llu) is not one these kernels reproduce",
                    threads_a, threads_b, (unsigned long long)offset_a, (unsigned long long)offset_b);
    int64_t lanes = int64_t(n) * res * res;
    if (int64_t(threads_a) > lanes) lanes = threads_a;
    if (draw_b && int64_t(threads_b) > lanes) lanes = threads_b;
    hipLaunchKernelGGL(rays_and_draws_kernel, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, as_stream(stream), cam2world, intrinsics, n, res, origins, dirs,
                       draw_a, numel_a, da, threads_a, draw_b, draw_b ? numel_b : 0, db, draw_b ? threads_b : 0u);
    return check_launch("make_rays_and_draws");
}

extern "C" int gnerf_torch_rand(float* out, int64_t numel, uint64_t seed, uint64_t offset, uint32_t threads, gnerf_stream_t stream) {
    using namespace gnerf;
    if (!out || numel < 1) return fail(GNERF_E_ARG, "torch_rand: null output or numel < 1");
    TorchRandDraw d;
    if (!torch_rand_draw(seed, offset, threads, numel, d))
        return fail(GNERF_E_UNSUPPORTED, "torch_rand: %u threads for %lld elements at offset %llu is not a geometry this kernel reproduces "
                    "(threads must be a power of two or >= numel, offset a multiple of 4)", threads, (long long)numel, (unsigned long long)offset);
    hipLaunchKernelGGL(torch_rand_kernel, dim3((unsigned)((numel + 255) / 256)), dim3(256), 0, as_stream(stream), out, numel, d);
    return check_launch("torch_rand");
}
