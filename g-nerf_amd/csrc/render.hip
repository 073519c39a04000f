// Fused tri-plane importance renderer for gfx950 (MI355X).
//
// One kernel does, per ray, everything ImportanceRenderer.forward does with a dozen whole-tensor PyTorch
// ops and ~4 GB of intermediates (reference training/volumetric_rendering/renderer.py:88-140,
// ray_marcher.py:25-57, triplane.py:113-136):
//
//   depth proposals -> tri-plane bilinear lookup -> 32->64->33 MLP -> coarse march -> importance
//   resampling -> fine lookup + MLP -> depth merge -> final composite.
//
// This file holds the pieces every kernel shares (depth proposals, importance resampling, the ray march, DPP scans),
// the GENERIC kernel described below (any sample count up to 256+256), the entry points and the kernel choice.  The
// kernels that run G-NeRF's actual configurations are in the .inl files included further down:
//   render_pipe.inl  3 shader waves + 1 scalar wave per workgroup, three rays in flight: up to 48+48 and up to 96+96 samples
//   render_coop.inl  3 waves per ray with barrier-separated phases (and the 16-sample shade tile all of them use)
//   render_bwd.inl   the backward pass (plane + decoder gradients) of the renderer and of run_model
//
// Generic kernel: ONE WAVE (a 64-lane workgroup) owns a ray at a time and walks a small tile of rays
// (4x4 pixels when the rays form an image).  Nothing but the three outputs is ever written to HBM.
//
//  * Lookup ("gather") layout: 8 adjacent lanes read one whole 128-byte texel (32 fp32 channels of the NHWC
//    planes) with one global_load_dwordx4 each, so a wave instruction touches 8 cache lines, the minimum.
//    A wave does 8 samples per step, two steps per 16-sample MLP tile.
//  * MLP on the matrix cores in exact fp32: v_mfma_f32_16x16x4_f32.  Layer 1 is computed transposed
//    (hidden on the accumulator rows, samples on the lanes) so that its accumulator registers are directly
//    the A operand of layer 2 (which sums over hidden) -- no data movement between the layers.  Layer 2
//    puts samples on accumulator rows and colour channels on lanes, which is what compositing wants
//    (a per-lane sum over registers).  The density row of layer 2 is a 16-term per-lane dot product plus
//    two cross-lane adds: a 33rd column would cost a third more MFMAs.
//  * Per-sample scalars (depth, density, weights, cdf, ranks) live in a few hundred bytes of LDS per wave;
//    colours of all samples (needed until the final weights are known) live in LDS as a lane-private
//    spill, 1 KiB per 16 samples.
//  * The call-wide depth clamp of ray_marcher.py:49-50 needs min/max over every depth of the call: waves
//    publish their extrema with two integer atomics, and a tiny second kernel applies the clamp.

#include "common.h"
#include "raygen.h"
#include <cstdlib>
#include <cstring>

namespace {

using namespace gnerf;

typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int kHidden = 64;
constexpr int kStagePitch = 36;        // dwords per staged sample row (32 channels + pad)
constexpr int kRaysPerWave = 16;

struct Params {
    gnerf_render_params p;
    float box_scale;        // 2 / box_warp
    float delta;            // (ray_end - ray_start) / (S - 1)
    float inv_start, inv_end, disp_delta;
    int tiles_c, tiles_f;   // 16-sample MLP tiles
    int n_tiles;            // ray tiles in the grid
    int tiles_per_item, tiles_y;   // image tiling (0 when rays are not an image).  tiles_per_item == 0 with tiles_y > 0 (set by the staged backward
                            // of RAGGED calls only: several items, rays per item not a multiple of 16): every item's rays are padded to tiles_y,
                            // a multiple of 16, in the ray SEQUENCE, so that no 16-ray tile straddles items -- linear_pad() below.  (Kept in a
                            // field the forward already has: one more kernel argument cost the headline kernel eight spilled SGPRs.)
    int total_rays;
    int split_shift;        // small launches: a 16-ray tile is shared by 1 << split_shift workgroups (coop / generic / backward kernels)
    int pipe_unit;          // pipelined kernel: rays dealt to a workgroup at a time (kPipeUnit; fewer for launches that do not fill the chip)
    unsigned tex_pitch, row_pitch, plane_pitch;     // byte addressing of a texel, see plane_taps (render_coop.inl)
    int64_t item_bytes;     // bytes from one item's planes to the next: 3 * H * W * 128, or 0 when every item reads the same planes (planes_shared)
    const float* absmax;    // GNERF_MLP_AUTO: max |planes| (one device float) for choose_mlp
    TorchRandDraw draw_c, draw_f;   // rng_mode: the two draws torch's generator would have made (raygen.h)
    uint64_t draw_item_ctr;         // rng_per_item: Philox counter stride from one item's draws to the next (offset stride / 4)
    // staged backward on the pipelined path: floats per ray of the exchange / staging buffer and from one 16-rank tile's block to the
    // next (33 n_all and 512 when the blocks are the dX rows of the staged scatter; n_all + 32 tiles and 32 for a decoder-only request)
    int64_t bwd_ray_stride;
    int bwd_tile_pitch;
};
__host__ __device__ __forceinline__ int linear_pad(const Params& P) { return P.tiles_per_item == 0 ? P.tiles_y : 0; }


// ---- order-preserving float <-> uint so that integer atomics give float min/max
__device__ __forceinline__ unsigned ord_encode(float f) {
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord_decode(unsigned u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

// min / max / clamp of values that are known not to be NaN.  Kernels run in IEEE mode, where fminf / fmaxf on an operand of unknown
// origin cost a canonicalising `v_max x, x, x` next to the compare (16 per 16-sample tile in the softplus).  clamp_nn is one
// v_med3_f32.  For min / max two ways around the canonicalisation were tried and measured: fmed3(a, b, +-inf) -- the compiler folds it
// straight back into fmaxf / fminf -- and the bare instruction as inline assembly -- 16 instructions fewer per tile (384 -> 368) and
// no change in kernel time (0.54 ms either way, inside the run-to-run spread).  So these are plain fminf / fmaxf.
__device__ __forceinline__ float max_nn(float a, float b) { return fmaxf(a, b); }
__device__ __forceinline__ float min_nn(float a, float b) { return fminf(a, b); }
__device__ __forceinline__ float clamp_nn(float x, float lo, float hi) { return __builtin_amdgcn_fmed3f(x, lo, hi); }

__device__ __forceinline__ float softplus_f(float x) {          // torch softplus, beta 1, threshold 20
    return x > 20.f ? x : log1pf(expf(x));
}
// MLP activations: these run once per hidden unit per sample, so they use the hardware exp2/log2.
__device__ __forceinline__ float softplus_fast(float x) {
    const float e = __expf(-fabsf(x));
    return fmaxf(x, 0.f) + __logf(1.f + e);
}
__device__ __forceinline__ float sigmoid_fast(float x) {
    return __frcp_rn(1.f + __expf(-x));
}

// ---- wave-wide scans on the DPP network (row_shr within 16-lane rows, then row_bcast:15 / :31 across rows).
// No LDS traffic and ALU latency only -- the ds_bpermute shuffles they replace cost an LDS round trip per step,
// and these scans sit on the per-ray critical path (transmittance product, cdf).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_mov(float old, float src) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src), CTRL, ROW_MASK, 0xf, false));
}
// Round 6: every step is ONE instruction that reads and writes the same register -- `v_op_dpp v, v, v <shift>`: lanes the shift gives
// no source (and rows the row_mask leaves out) are simply not written, i.e. keep their own value, which is the scan's identity step.
// Written through update_dpp + an arithmetic instruction the compiler can only fold the full-mask row shifts of the sum (old = 0 is
// its bound_ctrl zero); every row_bcast step and every step of the product scan came out as v_mov (old) + v_mov_dpp + op: 10 / 18
// vector instructions per scan instead of 6, on the wave whose per-ray pass is 476 of them (GNERF_DPP_INPLACE=0 is that form).
// The s_nop 1 are the two wait states a DPP read needs after the VALU write of its source; the compiler's hazard recogniser does not
// look inside (or behind) inline assembly, so the blocks begin and end with one as well.
#ifndef GNERF_DPP_INPLACE
#define GNERF_DPP_INPLACE 1
#endif
#define GNERF_SCAN6(OP, ZF)                                                                 \
    "s_nop 1\n\t"                                                                           \
    OP " %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" ZF "\n\ts_nop 1\n\t"              \
    OP " %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf" ZF "\n\ts_nop 1\n\t"              \
    OP " %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf" ZF "\n\ts_nop 1\n\t"              \
    OP " %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf" ZF "\n\ts_nop 1\n\t"              \
    OP " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"                 \
    OP " %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1"
__device__ __forceinline__ float wave_scan_add(float v, int /*lane*/) {         // inclusive
#if GNERF_DPP_INPLACE
    asm(GNERF_SCAN6("v_add_f32_dpp", " bound_ctrl:1") : "+v"(v));
#else
    v += dpp_mov<0x111, 0xf>(0.f, v);
    v += dpp_mov<0x112, 0xf>(0.f, v);
    v += dpp_mov<0x114, 0xf>(0.f, v);
    v += dpp_mov<0x118, 0xf>(0.f, v);
    v += dpp_mov<0x142, 0xa>(0.f, v);
    v += dpp_mov<0x143, 0xc>(0.f, v);
#endif
    return v;
}
__device__ __forceinline__ float wave_scan_mul(float v, int /*lane*/) {         // inclusive
#if GNERF_DPP_INPLACE
    asm(GNERF_SCAN6("v_mul_f32_dpp", "") : "+v"(v));
#else
    v *= dpp_mov<0x111, 0xf>(1.f, v);
    v *= dpp_mov<0x112, 0xf>(1.f, v);
    v *= dpp_mov<0x114, 0xf>(1.f, v);
    v *= dpp_mov<0x118, 0xf>(1.f, v);
    v *= dpp_mov<0x142, 0xa>(1.f, v);
    v *= dpp_mov<0x143, 0xc>(1.f, v);
#endif
    return v;
}
// two independent sums at once: the steps of one fill a wait state of the other
__device__ __forceinline__ void wave_scan_add2(float& a, float& b) {
#if GNERF_DPP_INPLACE
#define GNERF_STEP2(CTRL)                                                       \
    "v_add_f32_dpp %0, %0, %0 " CTRL "\n\tv_add_f32_dpp %1, %1, %1 " CTRL "\n\ts_nop 0\n\t"
    asm("s_nop 1\n\t"
        GNERF_STEP2("row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")
        GNERF_STEP2("row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1")
        GNERF_STEP2("row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1")
        GNERF_STEP2("row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1")
        GNERF_STEP2("row_bcast:15 row_mask:0xa bank_mask:0xf")
        GNERF_STEP2("row_bcast:31 row_mask:0xc bank_mask:0xf")
        "s_nop 0"
        : "+v"(a), "+v"(b));
#undef GNERF_STEP2
#else
    a = wave_scan_add(a, 0); b = wave_scan_add(b, 0);
#endif
}
#undef GNERF_SCAN6
__device__ __forceinline__ float wave_last(float v) {                            // lane 63's value, in every lane
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_shift_up(float v, float first) {           // lane i <- lane i-1, lane 0 <- first
    return dpp_mov<0x138, 0xf>(first, v);
}
__device__ __forceinline__ float wave_sum(float v) { return wave_last(wave_scan_add(v, 0)); }

// exp / softplus on the hardware exp2/log2 units (about 1 ulp each; absolute error ~1e-7 on the values used here)
__device__ __forceinline__ float exp_hw(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }
__device__ __forceinline__ float softplus_march(float x) {      // softplus with torch's threshold 20 (ray_marcher.py:33)
    const float e = __builtin_amdgcn_exp2f(-fabsf(x) * 1.44269504088896341f);
    const float sp = fmaf(__builtin_amdgcn_logf(1.0f + e), 0.693147180559945309f, fmaxf(x, 0.f));
    return x > 20.f ? x : sp;
}

// Per-wave LDS carve-up (all float/int, 16-byte aligned pieces)
struct Scratch {
    float* t_e;      // [S_pad]  element depths: coarse k at k, fine i at 16*tiles_c + i
    float* sig_e;    // [S_pad]  densities, same indexing
    float* v_e;      // [S_pad]  final colour weight of each element (0 for padding)
    int*   rank_e;   // [S_pad]
    float* s_t;      // [S_pad]  depths in sorted order
    float* s_sig;    // [S_pad]
    float* w_s;      // [S_pad]  weights (coarse march, then final march)
    float* cdf;      // [S_pad]
    float* stage;    // [16 * kStagePitch]
    float* colors;   // [(tiles_c + tiles_f) * 2 * 256]
};

__host__ __device__ inline size_t scratch_floats(int s_pad, int tiles) {
    return size_t(8) * s_pad + 16 * kStagePitch + size_t(tiles) * 512;
}

// The per-lane weight fragments (see the layout notes above).
struct Weights {
    float a1[4][8];     // layer 1 A operand: W1[16m + (lane&15)][8*(lane>>4) + s]
    float b1[4][4];     // layer 1 bias as accumulator init: b1[16m + 4*(lane>>4) + r]
    float w2s[4][4];    // density row of layer 2: W2[0][16m + 4*(lane>>4) + r]
    float b2w[2][16];   // layer 2 B operand: W2[1 + 16n + (lane&15)][16m + 4*(lane>>4) + r]  (index m*4+r)
    float b2c[2];       // b2[1 + 16n + (lane&15)]
    float b2s;          // b2[0]
};

__device__ __forceinline__ void load_weights(Weights& w, const gnerf_render_params& p, int lane) {
    const int j = lane & 15, g = lane >> 4;
#pragma unroll
    for (int m = 0; m < 4; m++) {
#pragma unroll
        for (int s = 0; s < 8; s++) w.a1[m][s] = p.w1[(16 * m + j) * 32 + 8 * g + s];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            w.b1[m][r] = p.b1[16 * m + 4 * g + r];
            w.w2s[m][r] = p.w2[16 * m + 4 * g + r];
        }
    }
#pragma unroll
    for (int n = 0; n < 2; n++) {
#pragma unroll
        for (int k = 0; k < 16; k++) w.b2w[n][k] = p.w2[(1 + 16 * n + j) * kHidden + 16 * (k >> 2) + 4 * g + (k & 3)];
        w.b2c[n] = p.b2[1 + 16 * n + j];
    }
    w.b2s = p.b2[0];
}

// Bilinear lookup of one plane for this lane's sample; adds (weight * 4 channels) into acc.
// u indexes W, v indexes H (grid_sample, align_corners=False, zero padding; renderer.py:64).
__device__ __forceinline__ void lookup_plane(v4f& acc, const float* __restrict__ plane, int H, int W, unsigned tex_q, unsigned row_q, float u, float v, int cq) {
    // tex_q / row_q: texel and row pitch in 16-byte units
    float ix = ((u + 1.f) * float(W) - 1.f) * 0.5f;
    float iy = ((v + 1.f) * float(H) - 1.f) * 0.5f;
    ix = fminf(fmaxf(ix, -1.5f), float(W) + 0.5f);     // keeps "everything out of range" out of range, and int conversion safe
    iy = fminf(fmaxf(iy, -1.5f), float(H) + 0.5f);
    const float x0f = floorf(ix), y0f = floorf(iy);
    const float fx = ix - x0f, fy = iy - y0f;
    const int x0 = int(x0f), y0 = int(y0f), x1 = x0 + 1, y1 = y0 + 1;
    const bool vx0 = x0 >= 0 && x0 < W, vx1 = x1 >= 0 && x1 < W, vy0 = y0 >= 0 && y0 < H, vy1 = y1 >= 0 && y1 < H;
    const int cx0 = min(max(x0, 0), W - 1), cx1 = min(max(x1, 0), W - 1);
    const int cy0 = min(max(y0, 0), H - 1), cy1 = min(max(y1, 0), H - 1);
    const float w00 = (vx0 && vy0) ? (1.f - fx) * (1.f - fy) : 0.f;
    const float w01 = (vx1 && vy0) ? fx * (1.f - fy) : 0.f;
    const float w10 = (vx0 && vy1) ? (1.f - fx) * fy : 0.f;
    const float w11 = (vx1 && vy1) ? fx * fy : 0.f;
    const v4f* base = reinterpret_cast<const v4f*>(plane) + cq;
    const v4f t00 = base[cy0 * row_q + cx0 * tex_q];
    const v4f t01 = base[cy0 * row_q + cx1 * tex_q];
    const v4f t10 = base[cy1 * row_q + cx0 * tex_q];
    const v4f t11 = base[cy1 * row_q + cx1 * tex_q];
    acc += t00 * w00 + t01 * w01 + t10 * w10 + t11 * w11;
}

// Shade `count` samples whose depths sit in lds.t_e[e0 ...]: densities -> lds.sig_e[e0 ...], colours ->
// lds.colors tiles tile0 ...
__device__ __forceinline__ void shade(const Params& P, const Weights& w, const Scratch& lds, const float* __restrict__ planes_item,
                                      float ox, float oy, float oz, float dx, float dy, float dz,
                                      int e0, int count, int ntiles, int tile0, int lane, const float* sig_noise = nullptr) {
    const int H = P.p.plane_h, W = P.p.plane_w;
    const int64_t plane_stride = P.plane_pitch / 4;
    const unsigned tex_q = P.tex_pitch / 16, row_q = P.row_pitch / 16;
    const int b = lane >> 3, cq = lane & 7;     // lookup layout: sample-in-step, channel quad
    const int j = lane & 15, g = lane >> 4;     // MFMA layout: sample-in-tile, k group
    for (int t = 0; t < ntiles; t++) {
        // ---- lookup: 2 steps x 8 samples, staged through LDS into the MFMA layout
#pragma unroll
        for (int a = 0; a < 2; a++) {
            const int js = 8 * a + b;
            const int idx = min(16 * t + js, count - 1);          // padding lanes re-shade the last sample
            const float depth = lds.t_e[e0 + idx];
            const float px = __fadd_rn(ox, __fmul_rn(depth, dx)) * P.box_scale;
            const float py = __fadd_rn(oy, __fmul_rn(depth, dy)) * P.box_scale;
            const float pz = __fadd_rn(oz, __fmul_rn(depth, dz)) * P.box_scale;
            v4f acc = {0.f, 0.f, 0.f, 0.f};
            lookup_plane(acc, planes_item, H, W, tex_q, row_q, px, py, cq);                          // plane 0: (x, y)
            lookup_plane(acc, planes_item + plane_stride, H, W, tex_q, row_q, px, pz, cq);           // plane 1: (x, z)
            lookup_plane(acc, planes_item + 2 * plane_stride, H, W, tex_q, row_q, pz, px, cq);       // plane 2: (z, x)
            acc *= (1.f / 3.f);                                                         // mean over planes, triplane.py:126
            *reinterpret_cast<v4f*>(lds.stage + js * kStagePitch + 4 * cq) = acc;
        }
        __syncthreads();
        const v4f f_lo = *reinterpret_cast<const v4f*>(lds.stage + j * kStagePitch + 8 * g);
        const v4f f_hi = *reinterpret_cast<const v4f*>(lds.stage + j * kStagePitch + 8 * g + 4);
        __syncthreads();
        const float f[8] = {f_lo[0], f_lo[1], f_lo[2], f_lo[3], f_hi[0], f_hi[1], f_hi[2], f_hi[3]};
        // ---- layer 1: H^T[64 x 16] = W1[64 x 32] . X^T[32 x 16], bias preloaded
        v4f h[4];
#pragma unroll
        for (int m = 0; m < 4; m++) h[m] = (v4f){w.b1[m][0], w.b1[m][1], w.b1[m][2], w.b1[m][3]};
#pragma unroll
        for (int s = 0; s < 8; s++) {
#pragma unroll
            for (int m = 0; m < 4; m++) h[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.a1[m][s], f[s], h[m], 0, 0, 0);
        }
        float sig = 0.f;
#pragma unroll
        for (int m = 0; m < 4; m++) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                h[m][r] = softplus_fast(h[m][r]);
                sig += w.w2s[m][r] * h[m][r];
            }
        }
        sig += __shfl_xor(sig, 16);
        sig += __shfl_xor(sig, 32);
        sig += w.b2s;
        if (g == 0 && 16 * t + j < count) lds.sig_e[e0 + 16 * t + j] = sig + (sig_noise ? sig_noise[16 * t + j] : 0.f);      // renderer.py:146-147
        // ---- layer 2: O[16 x 32] = H[16 x 64] . W2^T[64 x 32]
        v4f o[2];
#pragma unroll
        for (int n = 0; n < 2; n++) o[n] = (v4f){w.b2c[n], w.b2c[n], w.b2c[n], w.b2c[n]};
#pragma unroll
        for (int m = 0; m < 4; m++) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
#pragma unroll
                for (int n = 0; n < 2; n++) o[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(h[m][r], w.b2w[n][m * 4 + r], o[n], 0, 0, 0);
            }
        }
        // rgb = sigmoid(o) * 1.002 - 0.001 (triplane.py:134); lane (channel j of block n, samples 4g..4g+3)
#pragma unroll
        for (int n = 0; n < 2; n++) {
            v4f c;
#pragma unroll
            for (int r = 0; r < 4; r++) c[r] = sigmoid_fast(o[n][r]) * 1.002f - 0.001f;
            *reinterpret_cast<v4f*>(lds.colors + ((size_t(tile0 + t) * 2 + n) * 4 + g) * 64 + 4 * j) = c;
        }
    }
    __syncthreads();
}

// Stratified depth proposal k of a ray (renderer.py:169-192); bit-exact with torch's linspace + jitter.
__device__ __forceinline__ float coarse_depth(const Params& P, int64_t ray, int k) {
    const gnerf_render_params& p = P.p;
    const int S = p.depth_resolution;
    const float u = p.noise_coarse[ray * S + k];
    if (p.disparity_space_sampling) {
        const float step = 1.0f / float(S - 1);
        const float lin = (k < S / 2) ? __fmul_rn(step, float(k)) : __fsub_rn(1.0f, __fmul_rn(step, float(S - 1 - k)));
        const float q = __fadd_rn(lin, __fmul_rn(u, P.disp_delta));
        return __fdiv_rn(1.0f, __fadd_rn(__fmul_rn(P.inv_start, __fsub_rn(1.0f, q)), __fmul_rn(P.inv_end, q)));
    }
    if (p.ray_start_per_ray) {
        const float rs = p.ray_start_per_ray[ray], re = p.ray_end_per_ray[ray];
        const float span = __fsub_rn(re, rs);
        const float lin = __fadd_rn(rs, __fmul_rn(__fdiv_rn(float(k), float(S - 1)), span));      // math_utils.py:107-116
        return __fadd_rn(lin, __fmul_rn(u, __fdiv_rn(span, float(S - 1))));
    }
    const float step = __fdiv_rn(__fsub_rn(p.ray_end, p.ray_start), float(S - 1));            // torch.linspace
    const float lin = (k < S / 2) ? __fadd_rn(p.ray_start, __fmul_rn(step, float(k)))
                                  : __fsub_rn(p.ray_end, __fmul_rn(step, float(S - 1 - k)));
    return __fadd_rn(lin, __fmul_rn(u, P.delta));
}

// Importance resampling (renderer.py:194-253): fine depths from the coarse interval weights w[0..S-2].
// n_w = S-3 pdf entries, cdf has n_w+1.  `pdf_tmp` is scratch of >= S floats; `sync` separates the phases.
template <typename Sync>
__device__ __forceinline__ void resample_fine(const Params& P, int64_t ray, const float* t_c, const float* w, float* pdf_tmp, float* cdf,
                                              float* t_f, float* dbg, int n_all, int lane, Sync sync) {
    const gnerf_render_params& p = P.p;
    const int S = p.depth_resolution, F = p.depth_resolution_importance;
    const int n_w = S - 3;
    float part = 0.f;
    for (int i = lane; i < n_w; i += 64) {
        // smoothed weight a_{i+1} = (max(w_i, w_{i+1}) + max(w_{i+1}, w_{i+2})) / 2 + 0.01, then + 1e-5
        const float w0 = w[i], w1 = w[i + 1], w2 = w[i + 2];
        const float a = (fmaxf(w0, w1) + fmaxf(w1, w2)) * 0.5f + 0.01f;
        const float pw = a + 1e-5f;
        pdf_tmp[i] = pw;
        part += pw;
    }
    const float total = wave_sum(part);
    sync();
    float carry = 0.f;
    for (int base = 0; base < n_w; base += 64) {
        const int i = base + lane;
        const float pdf = (i < n_w) ? pdf_tmp[i] / total : 0.f;
        const float incl = wave_scan_add(pdf, lane) + carry;
        if (i < n_w) cdf[i + 1] = incl;
        carry = wave_last(incl);
    }
    if (lane == 0) cdf[0] = 0.f;
    sync();
    for (int i = lane; i < F; i += 64) {
        const float u = p.noise_fine[ray * F + i];
        // searchsorted(cdf[0..n_w], u, right=True): number of entries <= u
        int lo = 0, hi = n_w + 1;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (cdf[mid] <= u) lo = mid + 1; else hi = mid; }
        const int below = max(lo - 1, 0), above = min(lo, n_w);
        const float cb = cdf[below], ca = cdf[above];
        const float bb = (t_c[below] + t_c[below + 1]) * 0.5f;
        const float ba = (t_c[above] + t_c[above + 1]) * 0.5f;
        float denom = ca - cb;
        if (denom < 1e-5f) denom = 1.f;
        const float d = bb + (u - cb) / denom * (ba - bb);
        t_f[i] = d;
        if (dbg) dbg[GNERF_DBG_DEPTH_FINE * n_all + i] = d;
    }
    sync();
}

// Merge by depth (renderer.py:157-167): rank = number of elements ordered before this one, ties broken by
// position in cat([coarse, fine]) like a stable sort.  Writes rank_e and the sorted depth / density arrays.
__device__ __forceinline__ void merge_by_depth(const float* t_e, const float* sig_e, int* rank_e, float* s_t, float* s_sig,
                                               int S, int F, int fine_e0, int lane) {
    const int n_all = S + F;
    for (int q = lane; q < n_all; q += 64) {
        const int e = q < S ? q : fine_e0 + (q - S);
        const float key = t_e[e];
        int rank = 0;
        for (int o = 0; o < S; o++) { const float tk = t_e[o]; rank += (tk < key || (tk == key && o < q)) ? 1 : 0; }
        for (int o = 0; o < F; o++) { const float tk = t_e[fine_e0 + o]; rank += (tk < key || (tk == key && S + o < q)) ? 1 : 0; }
        rank_e[e] = rank;
        s_t[rank] = key;
        s_sig[rank] = sig_e[e];
    }
}

// Ray-march weights over n sorted samples (depth/density in LDS): writes w[0..n-2], returns sum(w) and
// sum(w * t_mid) in all lanes.  ray_marcher.py:26-42.
__device__ __forceinline__ void march(const float* t, const float* sig, float* w, int n, int lane, float& w_sum, float& wt_sum) {
    float carry = 1.f, acc_w = 0.f, acc_wt = 0.f;
    for (int base = 0; base < n - 1; base += 64) {
        const int k = base + lane;
        const bool ok = k < n - 1;
        float alpha = 0.f, tmid = 0.f;
        if (ok) {
            const float t0 = t[k], t1 = t[k + 1];
            const float delta = t1 - t0;
            const float smid = softplus_march((sig[k] + sig[k + 1]) * 0.5f - 1.f);
            tmid = (t0 + t1) * 0.5f;
            alpha = 1.f - exp_hw(-(smid * delta));
        }
        const float x = ok ? (1.f - alpha + 1e-10f) : 1.f;
        const float incl = wave_scan_mul(x, lane);
        const float excl = wave_shift_up(incl, 1.f);
        const float wk = alpha * (excl * carry);
        carry *= wave_last(incl);
        if (ok) { w[k] = wk; acc_w += wk; acc_wt += wk * tmid; }
    }
    wave_scan_add2(acc_w, acc_wt);
    w_sum = wave_last(acc_w);
    wt_sum = wave_last(acc_wt);
}

// ---- the call-wide depth range (ray_marcher.py:49-50).
// Workspace words: [0] ~ord_encode(min depth)  [1] ord_encode(max depth)  [3] clamp blocks finished.
// ALL ZERO means idle ("min = +inf, max = -inf"): the caller provides a zeroed workspace once and every call leaves it
// zeroed (clamp_depth_kernel's last block), so no launch is spent on initialising it.
// (Also tried: letting the last render workgroup to finish apply the clamp itself for small launches, to save the second
// launch -- the device-scope fences that needs at the end of every workgroup cost more than the launch: 86 -> 118 us per
// 64x64-ray frame.)
// depth_clamp_per_item: one (min, max) pair per item at words [16 + 2 item], [17 + 2 item] instead of the call-wide pair.
constexpr int kClampItemWord0 = 16, kClampMaxItems = 4096;
__device__ __forceinline__ void publish_depth_range(const Params& P, float blk_min, float blk_max, int item = 0) {
    unsigned* ws = static_cast<unsigned*>(P.p.workspace) + (P.p.depth_clamp_per_item ? kClampItemWord0 + 2 * item : 0);
    atomicMax(ws + 0, ~ord_encode(blk_min));
    atomicMax(ws + 1, ord_encode(blk_max));
}
// A workgroup's running depth range, published per item when the clamp is per item (a workgroup's rays may span items).
struct DepthRange {
    float mn = INFINITY, mx = -INFINITY;
    int item = -1;
    __device__ __forceinline__ void add(const Params& P, int it, float lo, float hi) {
        if (P.p.depth_clamp_per_item && it != item) { flush(P); item = it; }
        mn = fminf(mn, lo); mx = fmaxf(mx, hi);
    }
    __device__ __forceinline__ void flush(const Params& P) {
        if (mn <= mx) publish_depth_range(P, mn, mx, item < 0 ? 0 : item);
        mn = INFINITY; mx = -INFINITY;
    }
};

__global__ __launch_bounds__(64, 2) void render_kernel_generic(Params P) {
    extern __shared__ __align__(16) float smem[];
    const gnerf_render_params& p = P.p;
    const int lane = threadIdx.x;
    const int S = p.depth_resolution, F = p.depth_resolution_importance;
    const int s_pad = 16 * (P.tiles_c + P.tiles_f);
    Scratch lds;
    lds.t_e = smem;
    lds.sig_e = lds.t_e + s_pad;
    lds.v_e = lds.sig_e + s_pad;
    lds.rank_e = reinterpret_cast<int*>(lds.v_e + s_pad);
    lds.s_t = lds.v_e + 2 * s_pad;
    lds.s_sig = lds.s_t + s_pad;
    lds.w_s = lds.s_sig + s_pad;
    lds.cdf = lds.w_s + s_pad;
    lds.stage = lds.cdf + s_pad;
    lds.colors = lds.stage + 16 * kStagePitch;

    // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs, so give each XCD a
    // contiguous run of ray tiles (its L2 then sees neighbouring rays).  Speed only.
    const int n_groups = P.n_tiles << P.split_shift;
    const int per_xcd = (n_groups + kNumXCD - 1) / kNumXCD;
    const int group = (blockIdx.x % kNumXCD) * per_xcd + blockIdx.x / kNumXCD;
    const int tile = group >> P.split_shift;
    const int rr_count = group < n_groups ? (kRaysPerWave >> P.split_shift) : 0;        // surplus workgroups of the rounded-up grid only report in
    const int rr_first = (group & ((1 << P.split_shift) - 1)) * (kRaysPerWave >> P.split_shift);

    Weights w;
    load_weights(w, p, lane);
    const int fine_e0 = 16 * P.tiles_c;
    const int n_all = S + F;
    DepthRange range;

    for (int rr = rr_first; rr < rr_first + rr_count; rr++) {
        // ---- which ray
        int64_t ray;
        if (P.tiles_per_item > 0) {             // 4x4 pixel tiles, walked down image columns
            const int item = tile / P.tiles_per_item, tt = tile % P.tiles_per_item;
            const int tx = tt / P.tiles_y, ty = tt % P.tiles_y;
            const int x = tx * 4 + (rr & 3), y = ty * 4 + (rr >> 2);
            ray = int64_t(item) * p.rays_per_item + int64_t(y) * p.image_width + x;
        } else {
            ray = int64_t(tile) * kRaysPerWave + rr;
            if (ray >= P.total_rays) break;
        }
        const int item = int(ray / p.rays_per_item);
        const float ox = p.ray_origins[ray * 3 + 0], oy = p.ray_origins[ray * 3 + 1], oz = p.ray_origins[ray * 3 + 2];
        const float dx = p.ray_dirs[ray * 3 + 0], dy = p.ray_dirs[ray * 3 + 1], dz = p.ray_dirs[ray * 3 + 2];
        const float* planes_item = reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.planes_nhwc) + int64_t(item) * P.item_bytes);
        float* dbg = p.debug ? p.debug + ray * GNERF_DEBUG_SLOTS * n_all : nullptr;

        // ---- stratified depth proposals (renderer.py:169-192)
        for (int k = lane; k < S; k += 64) {
            const float d = coarse_depth(P, ray, k);
            lds.t_e[k] = d;
            if (dbg) dbg[GNERF_DBG_DEPTH_COARSE * n_all + k] = d;
        }
        for (int k = lane; k < s_pad; k += 64) lds.v_e[k] = 0.f;
        __syncthreads();

        // ---- coarse pass
        shade(P, w, lds, planes_item, ox, oy, oz, dx, dy, dz, 0, S, P.tiles_c, 0, lane, p.sigma_noise_coarse ? p.sigma_noise_coarse + ray * S : nullptr);
        if (dbg) for (int k = lane; k < S; k += 64) dbg[GNERF_DBG_SIGMA_COARSE * n_all + k] = lds.sig_e[k];

        float w_sum, wt_sum;
        if (F > 0) {
            march(lds.t_e, lds.sig_e, lds.w_s, S, lane, w_sum, wt_sum);
            __syncthreads();
            if (dbg) for (int k = lane; k < S - 1; k += 64) dbg[GNERF_DBG_WEIGHT_COARSE * n_all + k] = lds.w_s[k];

            resample_fine(P, ray, lds.t_e, lds.w_s, lds.s_sig, lds.cdf, lds.t_e + fine_e0, dbg, n_all, lane, [] { __syncthreads(); });

            // ---- fine pass
            shade(P, w, lds, planes_item, ox, oy, oz, dx, dy, dz, fine_e0, F, P.tiles_f, P.tiles_c, lane, p.sigma_noise_fine ? p.sigma_noise_fine + ray * F : nullptr);
            if (dbg) for (int k = lane; k < F; k += 64) dbg[GNERF_DBG_SIGMA_FINE * n_all + k] = lds.sig_e[fine_e0 + k];

            merge_by_depth(lds.t_e, lds.sig_e, lds.rank_e, lds.s_t, lds.s_sig, S, F, fine_e0, lane);
            __syncthreads();
            march(lds.s_t, lds.s_sig, lds.w_s, n_all, lane, w_sum, wt_sum);
            __syncthreads();
            // colour weight of the sample at sorted position r: (w[r-1] + w[r]) / 2  (midpoint colours, ray_marcher.py:27)
            for (int q = lane; q < n_all; q += 64) {
                const int e = q < S ? q : fine_e0 + (q - S);
                const int r = lds.rank_e[e];
                const float wl = r > 0 ? lds.w_s[r - 1] : 0.f, wr = r < n_all - 1 ? lds.w_s[r] : 0.f;
                lds.v_e[e] = (wl + wr) * 0.5f;
            }
            if (dbg) {
                for (int k = lane; k < n_all; k += 64) { dbg[GNERF_DBG_DEPTH_SORTED * n_all + k] = lds.s_t[k]; dbg[GNERF_DBG_SIGMA_SORTED * n_all + k] = lds.s_sig[k]; }
                for (int k = lane; k < n_all - 1; k += 64) dbg[GNERF_DBG_WEIGHT_FINAL * n_all + k] = lds.w_s[k];
            }
            range.add(P, item, lds.s_t[0], lds.s_t[n_all - 1]);
        } else {
            march(lds.t_e, lds.sig_e, lds.w_s, S, lane, w_sum, wt_sum);
            __syncthreads();
            for (int k = lane; k < S; k += 64) {
                const float wl = k > 0 ? lds.w_s[k - 1] : 0.f, wr = k < S - 1 ? lds.w_s[k] : 0.f;
                lds.v_e[k] = (wl + wr) * 0.5f;
            }
            if (dbg) for (int k = lane; k < S - 1; k += 64) dbg[GNERF_DBG_WEIGHT_FINAL * n_all + k] = lds.w_s[k];
            // the reference takes min/max over the depths tensor; coarse depths ascend (up to rounding ties)
            float mn = INFINITY, mx = -INFINITY;
            for (int k = lane; k < S; k += 64) { mn = fminf(mn, lds.t_e[k]); mx = fmaxf(mx, lds.t_e[k]); }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { mn = fminf(mn, __shfl_xor(mn, o)); mx = fmaxf(mx, __shfl_xor(mx, o)); }
            range.add(P, item, mn, mx);
        }
        __syncthreads();

        // ---- colours: rgb[c] = sum_e v_e * colour_e[c]; lane = (channel j of block n, samples 4g..4g+3 of each tile)
        const int j = lane & 15, g = lane >> 4;
        float acc[2] = {0.f, 0.f};
        const int ntile_all = P.tiles_c + P.tiles_f;
        for (int t = 0; t < ntile_all; t++) {
            const v4f v = *reinterpret_cast<const v4f*>(lds.v_e + 16 * t + 4 * g);
#pragma unroll
            for (int n = 0; n < 2; n++) {
                const v4f c = *reinterpret_cast<const v4f*>(lds.colors + ((size_t(t) * 2 + n) * 4 + g) * 64 + 4 * j);
                acc[n] += v[0] * c[0] + v[1] * c[1] + v[2] * c[2] + v[3] * c[3];
            }
        }
#pragma unroll
        for (int n = 0; n < 2; n++) {
            acc[n] += __shfl_xor(acc[n], 16);
            acc[n] += __shfl_xor(acc[n], 32);
            if (p.white_back) acc[n] = acc[n] + 1.f - w_sum;
            acc[n] = acc[n] * 2.f - 1.f;
        }
        if (g < 2) p.out_rgb[ray * 32 + 16 * g + j] = g == 0 ? acc[0] : acc[1];
        if (lane == 0) {
            float depth = wt_sum / w_sum;
            if (depth != depth) depth = INFINITY;        // nan_to_num(nan=inf), ray_marcher.py:49; clamp applied by clamp_depth_kernel
            p.out_depth[ray] = depth;
            p.out_wsum[ray] = w_sum;
        }
        __syncthreads();
    }
    if (lane == 0) range.flush(P);
}

// The clamp.  The last block to finish puts the workspace back to idle.
constexpr int kClampPerThread = 4;
__global__ __launch_bounds__(256) void clamp_depth_kernel(float* depth, unsigned* ws, int64_t n, int rays_per_item, int n_items_per_item_clamp) {
    const bool per_item = n_items_per_item_clamp > 0;
    float lo = 0.f, hi = 0.f;
    if (!per_item) { lo = ord_decode(~ws[0]); hi = ord_decode(ws[1]); }
#pragma unroll
    for (int k = 0; k < kClampPerThread; k++) {
        const int64_t i = (int64_t(blockIdx.x) * kClampPerThread + k) * 256 + threadIdx.x;
        if (i < n) {
            if (per_item) { const unsigned* w = ws + kClampItemWord0 + 2 * int(i / rays_per_item); lo = ord_decode(~w[0]); hi = ord_decode(w[1]); }
            depth[i] = fminf(fmaxf(depth[i], lo), hi);   // torch.clamp(x, min, max)
        }
    }
    __syncthreads();                                             // every thread of the block has read (and used) the range
    __shared__ int last;
    if (threadIdx.x == 0) last = atomicAdd(ws + 3, 1u) == gridDim.x - 1;
    __syncthreads();
    if (last) {                                                  // the last block puts the workspace back to idle
        if (threadIdx.x == 0) { ws[0] = 0u; ws[1] = 0u; ws[3] = 0u; }
        for (int i = threadIdx.x; i < 2 * n_items_per_item_clamp; i += 256) ws[kClampItemWord0 + i] = 0u;
    }
}

// ---------------------------------------------------------------------------------------------
// run_model for arbitrary points (renderer.py:142-148): same lookup + MLP, 16 points per step.

__global__ __launch_bounds__(64) void query_kernel(gnerf_render_params p, float box_scale, int n_points, int n_tiles,
                                                   const float* __restrict__ points, float* __restrict__ out_sigma, float* __restrict__ out_rgb) {
    __shared__ __align__(16) float stage[16 * kStagePitch];
    const int lane = threadIdx.x;
    const int H = p.plane_h, W = p.plane_w;
    const int64_t item_stride = int64_t(3) * H * W * 32;
    const int64_t plane_stride = p.planes_interleaved ? 32 : int64_t(H) * W * 32;
    const unsigned tex_q = p.planes_interleaved ? 24 : 8, row_q = tex_q * unsigned(W);
    const int b = lane >> 3, cq = lane & 7, j = lane & 15, g = lane >> 4;
    Weights w;
    load_weights(w, p, lane);
    const int tiles_per_item = (n_points + 15) / 16;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int item = tile / tiles_per_item, t = tile % tiles_per_item;
        const float* planes_item = p.planes_nhwc + int64_t(item) * item_stride;
        const float* pts = points + int64_t(item) * n_points * 3;
#pragma unroll
        for (int a = 0; a < 2; a++) {
            const int js = 8 * a + b;
            const int idx = min(16 * t + js, n_points - 1);
            const float px = pts[idx * 3 + 0] * box_scale, py = pts[idx * 3 + 1] * box_scale, pz = pts[idx * 3 + 2] * box_scale;
            v4f acc = {0.f, 0.f, 0.f, 0.f};
            lookup_plane(acc, planes_item, H, W, tex_q, row_q, px, py, cq);
            lookup_plane(acc, planes_item + plane_stride, H, W, tex_q, row_q, px, pz, cq);
            lookup_plane(acc, planes_item + 2 * plane_stride, H, W, tex_q, row_q, pz, px, cq);
            acc *= (1.f / 3.f);
            *reinterpret_cast<v4f*>(stage + js * kStagePitch + 4 * cq) = acc;
        }
        __syncthreads();
        const v4f f_lo = *reinterpret_cast<const v4f*>(stage + j * kStagePitch + 8 * g);
        const v4f f_hi = *reinterpret_cast<const v4f*>(stage + j * kStagePitch + 8 * g + 4);
        __syncthreads();
        const float f[8] = {f_lo[0], f_lo[1], f_lo[2], f_lo[3], f_hi[0], f_hi[1], f_hi[2], f_hi[3]};
        v4f h[4];
#pragma unroll
        for (int m = 0; m < 4; m++) h[m] = (v4f){w.b1[m][0], w.b1[m][1], w.b1[m][2], w.b1[m][3]};
#pragma unroll
        for (int s = 0; s < 8; s++) {
#pragma unroll
            for (int m = 0; m < 4; m++) h[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.a1[m][s], f[s], h[m], 0, 0, 0);
        }
        float sig = 0.f;
#pragma unroll
        for (int m = 0; m < 4; m++) {
#pragma unroll
            for (int r = 0; r < 4; r++) { h[m][r] = softplus_fast(h[m][r]); sig += w.w2s[m][r] * h[m][r]; }
        }
        sig += __shfl_xor(sig, 16);
        sig += __shfl_xor(sig, 32);
        sig += w.b2s;
        const int pt = 16 * t + j;
        if (g == 0 && pt < n_points) out_sigma[int64_t(item) * n_points + pt] = sig;
        if (!out_rgb) continue;                 // densities only (shape extraction): the colour half of layer 2 is not evaluated
        v4f o[2];
#pragma unroll
        for (int n = 0; n < 2; n++) o[n] = (v4f){w.b2c[n], w.b2c[n], w.b2c[n], w.b2c[n]};
#pragma unroll
        for (int m = 0; m < 4; m++) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
#pragma unroll
                for (int n = 0; n < 2; n++) o[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(h[m][r], w.b2w[n][m * 4 + r], o[n], 0, 0, 0);
            }
        }
#pragma unroll
        for (int n = 0; n < 2; n++) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int q = 16 * t + 4 * g + r;
                if (q < n_points) out_rgb[(int64_t(item) * n_points + q) * 32 + 16 * n + j] = sigmoid_fast(o[n][r]) * 1.002f - 0.001f;
            }
        }
    }
}

// ---- GNERF_MLP_AUTO: which decoder arithmetic may this call use?  Evaluated by every workgroup of the pipe / coop kernels before
// it stages the decoder (one wave, a few hundred loads from L2, ~0.1 % of a workgroup's work): no separate launch.
// The f16 hi/lo split (render_coop.inl) represents an operand v as hi + lo with |v - hi - lo| <= max(2^-22 |v|, 2^-25): fp32-grade
// for operands well inside f16's range, but an ABSOLUTE 2^-25 per operand once the low half goes subnormal, and inf/NaN once the high
// half overflows.  With A = max |planes| (the features are convex combinations of texels, so |x| <= A), W1' = log2(e) W1 and the
// row norms R1 = max_r ||W1'[r,:]||_2, R2 = max_r ||W2[r,:]||_2 the decision is
//   no overflow (hard bounds):  A, max |W1'|, max |W2|, and  max_r ||W1'[r,:]||_1 A + max |b1'| + 1  (>= every hidden activation) <= 30000
//   accuracy (random-sign error model, the same one that gives the fp32 reference its ~2^-24 sqrt(n) behaviour):
//     e_p = 2^-22 R1 A + 2^-25 (R1 + sqrt(32) A)                       error of a layer-1 pre-activation (base-2 scaled)
//     h   = R1 A + max |b1'| + 1                                        scale of the hidden activations
//     e_o = R2 e_p + 2^-22 R2 h + 2^-25 (R2 + 8 h)                      error of a layer-2 output
//   f16x3 iff e_o <= 2^-12: |d rgb| <= 0.25 ln2 e_o = 4e-5 in the worst case (pixel MSE < 2e-9), typically 1e-3 of that.
// Anything else -- including NaN/inf anywhere in the operands -- takes the exact-fp32 arithmetic.  BASELINE config 2 (randn planes,
// default-init decoder) has e_o = 1.1e-5; planes scaled by ~20 or decoder weights by ~5 cross over to fp32.
constexpr float kMlpRangeLimit = 30000.f;
constexpr float kMlpErrLimit = 1.0f / 4096.f;
// ... and the f16 body's softplus may take its short form log2(1 + 2^p') -- exp2, add, log2 instead of exp2, add, log2, add, max per
// hidden activation -- while no base-2 pre-activation can reach exp2's overflow: |p'| <= max_r ||W1'[r,:]||_1 A + max |b1'| <= 99
// (config 2: 37; on a random-init generator's planes: 57).  Beyond that the body keeps the form that is safe for any p'.
constexpr float kSoftplusDirectLimit = 100.f;
__device__ __forceinline__ int choose_mlp(const Params& P, float* smem, bool* softplus_direct = nullptr) {
    const gnerf_render_params& p = P.p;
    constexpr float kL2e = 1.44269504088896341f;
    // the decoder into LDS with coalesced loads (rows padded to an odd pitch: the row sums below are conflict-free); a first
    // version read the rows straight from global memory, one row per lane -- 96 uncoalesced load instructions in front of every
    // workgroup's first ray cost 40 us per launch
    constexpr int kP1 = 33, kP2 = 65;
    float* s1 = smem + 4;
    float* s2 = s1 + 64 * kP1;
    for (int i = threadIdx.x; i < 64 * 32; i += blockDim.x) s1[(i >> 5) * kP1 + (i & 31)] = p.w1[i];
    for (int i = threadIdx.x; i < 33 * 64; i += blockDim.x) s2[(i >> 6) * kP2 + (i & 63)] = p.w2[i];
    __syncthreads();
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        // lane r < 64: row r of W1; lane r < 33: row r of W2
        float l1 = 0.f, sq1 = 0.f, mx1 = 0.f;
#pragma unroll 8
        for (int c = 0; c < 32; c++) { const float w = fabsf(s1[lane * kP1 + c]) * kL2e; l1 += w; sq1 = fmaf(w, w, sq1); mx1 = fmaxf(mx1, w); }
        float sq2 = 0.f, mx2 = 0.f;
        if (lane < 33) {
#pragma unroll 8
            for (int c = 0; c < 64; c++) { const float w = fabsf(s2[lane * kP2 + c]); sq2 = fmaf(w, w, sq2); mx2 = fmaxf(mx2, w); }
        }
        float mb1 = fabsf(p.b1[lane]) * kL2e;
        float mb2 = lane < 33 ? fabsf(p.b2[lane]) * kL2e : 0.f;
        // NaN-propagating maxima: fmaxf drops NaNs, so carry "anything not finite" separately
        bool bad = !(l1 < INFINITY) || !(sq2 < INFINITY) || !(mb1 < INFINITY) || !(mb2 < INFINITY);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            l1 = fmaxf(l1, __shfl_xor(l1, o)); sq1 = fmaxf(sq1, __shfl_xor(sq1, o)); mx1 = fmaxf(mx1, __shfl_xor(mx1, o));
            sq2 = fmaxf(sq2, __shfl_xor(sq2, o)); mx2 = fmaxf(mx2, __shfl_xor(mx2, o));
            mb1 = fmaxf(mb1, __shfl_xor(mb1, o)); mb2 = fmaxf(mb2, __shfl_xor(mb2, o));
        }
        bad = __any(bad);
        if (lane == 0) {
            const float A = *P.absmax;
            const float R1 = sqrtf(sq1), R2 = sqrtf(sq2);
            const float h_hard = l1 * A + mb1 + 1.f;
            const float e_p = 0x1p-22f * R1 * A + 0x1p-25f * (R1 + 5.657f * A);
            const float h = R1 * A + mb1 + 1.f;
            const float e_o = R2 * e_p + 0x1p-22f * R2 * h + 0x1p-25f * (R2 + 8.f * h);
            const bool ok = !bad && A <= kMlpRangeLimit && mx1 <= kMlpRangeLimit && mx2 <= kMlpRangeLimit && h_hard <= kMlpRangeLimit
                            && mb2 <= kMlpRangeLimit && e_o <= kMlpErrLimit;         // every comparison is false for NaN
            const int choice = ok ? 1 : 2;                                           // kMlpF16x3 : kMlpF32
            reinterpret_cast<int*>(smem)[0] = choice;
            reinterpret_cast<int*>(smem)[1] = (ok && h_hard <= kSoftplusDirectLimit) ? 1 : 0;
            if (blockIdx.x == 0 && p.workspace) static_cast<int*>(p.workspace)[4] = choice;         // diagnostics (gnerf_hip.last_mlp_choice)
        }
    }
    __syncthreads();
    const int choice = __builtin_amdgcn_readfirstlane(reinterpret_cast<const int*>(smem)[0]);
    if (softplus_direct) *softplus_direct = __builtin_amdgcn_readfirstlane(reinterpret_cast<const int*>(smem)[1]) != 0;
    __syncthreads();                                     // smem is the body's from here on
    return choice;
}

#include "render_coop.inl"
#include "render_pipe.inl"
#include "render_bwd.inl"

// (PerDeviceOnce: csrc/common.h)

int check_common(const gnerf_render_params* p) {
    if (!p) return fail(GNERF_E_ARG, "render: params is null");
    if (!p->planes_nhwc || !p->w1 || !p->b1 || !p->w2 || !p->b2) return fail(GNERF_E_ARG, "render: planes and decoder weights must not be null");
    if (p->n_items < 1 || p->plane_h < 1 || p->plane_w < 1) return fail(GNERF_E_ARG, "render: bad plane shape");
    if (int64_t(p->planes_shared ? 1 : p->n_items) * 3 * p->plane_h * p->plane_w * 32 > INT32_MAX * int64_t(4))
        return fail(GNERF_E_ARG, "render: planes too large");
    if (!(p->box_warp > 0.f)) return fail(GNERF_E_ARG, "render: box_warp must be positive");
    if (p->planes_interleaved != 0 && p->planes_interleaved != 1) return fail(GNERF_E_ARG, "render: planes_interleaved must be 0 or 1");
    return GNERF_OK;
}

void fill_pitches(Params& P) {
    const unsigned W = unsigned(P.p.plane_w), H = unsigned(P.p.plane_h);
    P.tex_pitch = P.p.planes_interleaved ? 384u : 128u;
    P.row_pitch = P.tex_pitch * W;
    P.plane_pitch = P.p.planes_interleaved ? 128u : H * W * 128u;
}

#include "scatter_binned.inl"

}  // namespace

extern "C" size_t gnerf_render_workspace_bytes(void) { return size_t(kClampItemWord0 + 2 * kClampMaxItems) * 4; }

// Validation and derived launch parameters shared by the forward and backward entry points.
static int fill_params(const gnerf_render_params* p, Params& P) {
    using namespace gnerf;
    if (int e = check_common(p)) return e;
    const bool own_rays = !p->ray_origins && !p->ray_dirs && p->cam2world && p->intrinsics;
    if (!own_rays && (!p->ray_origins || !p->ray_dirs)) return fail(GNERF_E_ARG, "render: rays must not be null (or both null with cam2world and intrinsics given)");
    // ray tensors AND cameras: the generating instantiation keys on cam2world and would replace the caller's rays by camera rays
    // (and divide by image_width, which nothing has checked for this combination): one source of rays per call
    if (!own_rays && (p->cam2world || p->intrinsics)) return fail(GNERF_E_ARG, "render: give ray tensors or cam2world / intrinsics, not both");
    if (own_rays && (p->image_width < 1 || int64_t(p->image_width) * p->image_width != p->rays_per_item))
        return fail(GNERF_E_ARG, "render: in-kernel rays need rays_per_item = image_width^2");
    if (p->rng_mode != GNERF_RNG_TENSORS && p->rng_mode != GNERF_RNG_TORCH_PHILOX) return fail(GNERF_E_ARG, "render: rng_mode %d is not one of GNERF_RNG_*", p->rng_mode);
    if (p->rng_mode == GNERF_RNG_TENSORS && !p->noise_coarse) return fail(GNERF_E_ARG, "render: noise_coarse must not be null");
    if (p->rng_mode == GNERF_RNG_TORCH_PHILOX && (p->noise_coarse || p->noise_fine)) return fail(GNERF_E_ARG, "render: rng_mode = GNERF_RNG_TORCH_PHILOX takes no noise tensors");
    const int S = p->depth_resolution, F = p->depth_resolution_importance;
    if (S < 2 || S > GNERF_MAX_SAMPLES) return fail(GNERF_E_ARG, "render: depth_resolution %d outside [2, %d]", S, GNERF_MAX_SAMPLES);
    if (F < 0 || F > GNERF_MAX_SAMPLES) return fail(GNERF_E_ARG, "render: depth_resolution_importance %d outside [0, %d]", F, GNERF_MAX_SAMPLES);
    if (F > 0 && S < 4) return fail(GNERF_E_ARG, "render: importance sampling needs depth_resolution >= 4");
    if (F > 0 && !p->noise_fine && p->rng_mode == GNERF_RNG_TENSORS) return fail(GNERF_E_ARG, "render: noise_fine is null but depth_resolution_importance > 0");
    if (p->sigma_noise_fine && !p->sigma_noise_coarse) return fail(GNERF_E_ARG, "render: sigma_noise_fine without sigma_noise_coarse");
    if (p->sigma_noise_coarse && F > 0 && !p->sigma_noise_fine) return fail(GNERF_E_ARG, "render: sigma_noise_coarse is given but sigma_noise_fine is null with depth_resolution_importance > 0");
    if (p->sigma_noise_coarse && (own_rays || p->rng_mode != GNERF_RNG_TENSORS)) return fail(GNERF_E_UNSUPPORTED, "render: density noise comes with ray and noise tensors (not with in-kernel rays / draws)");
    if (p->rays_per_item < 1) return fail(GNERF_E_ARG, "render: rays_per_item must be positive");
    if ((p->ray_start_per_ray == nullptr) != (p->ray_end_per_ray == nullptr)) return fail(GNERF_E_ARG, "render: per-ray start and end must be given together");
    const int64_t total = int64_t(p->n_items) * p->rays_per_item;
    if (total * (S + F) > INT32_MAX * int64_t(8)) return fail(GNERF_E_ARG, "render: too many samples in one call");

    P.p = *p;
    fill_pitches(P);
    if ((p->planes_shared != 0 && p->planes_shared != 1) || (p->depth_clamp_per_item != 0 && p->depth_clamp_per_item != 1))
        return fail(GNERF_E_ARG, "render: planes_shared and depth_clamp_per_item must be 0 or 1");
    if (p->depth_clamp_per_item && p->n_items > kClampMaxItems) return fail(GNERF_E_ARG, "render: depth_clamp_per_item supports at most %d items", kClampMaxItems);
    P.item_bytes = p->planes_shared ? 0 : int64_t(3) * p->plane_h * p->plane_w * 128;
    if (P.row_pitch >= (1u << 24)) return fail(GNERF_E_UNSUPPORTED, "render: plane rows wider than 2^24 bytes");
    P.box_scale = float(2.0 / double(p->box_warp));
    P.delta = float((double(p->ray_end) - double(p->ray_start)) / double(S - 1));
    P.inv_start = float(1.0 / double(p->ray_start));
    P.inv_end = float(1.0 / double(p->ray_end));
    P.disp_delta = float(1.0 / double(S - 1));
    P.tiles_c = (S + 15) / 16;
    P.tiles_f = (F + 15) / 16;
    P.total_rays = int(total);
    P.split_shift = 0;
    P.pipe_unit = 8;
    P.absmax = nullptr;
    P.draw_c = P.draw_f = TorchRandDraw{};
    P.draw_item_ctr = 0;
    if (p->rng_mode == GNERF_RNG_TORCH_PHILOX) {
        const int64_t units = p->rng_per_item ? int64_t(p->rays_per_item) : total;       // rays one draw covers
        if ((p->rng_per_item != 0 && p->rng_per_item != 1) || p->rng_offset_item_stride % 4 != 0)
            return fail(GNERF_E_ARG, "render: rng_per_item must be 0 or 1 and rng_offset_item_stride a multiple of 4");
        if (!torch_rand_draw(p->rng_seed, p->rng_offset_coarse, p->rng_threads_coarse, units * S, P.draw_c) ||
            (F > 0 && !torch_rand_draw(p->rng_seed, p->rng_offset_fine, p->rng_threads_fine, units * F, P.draw_f)))
            return fail(GNERF_E_UNSUPPORTED, "render: the generator geometry (threads %u / %u, offsets %llu / %llu) is not one the in-kernel draws reproduce",
                        p->rng_threads_coarse, p->rng_threads_fine, (unsigned long long)p->rng_offset_coarse, (unsigned long long)p->rng_offset_fine);
        P.draw_item_ctr = p->rng_per_item ? p->rng_offset_item_stride / 4 : 0;
    }
    const int iw = p->image_width;
    if (iw > 0 && iw % 4 == 0 && p->rays_per_item % iw == 0 && (p->rays_per_item / iw) % 4 == 0) {
        P.tiles_y = p->rays_per_item / iw / 4;
        P.tiles_per_item = (iw / 4) * P.tiles_y;
        P.n_tiles = P.tiles_per_item * p->n_items;
    } else {
        P.tiles_y = 0;
        P.tiles_per_item = 0;
        P.n_tiles = int((total + kRaysPerWave - 1) / kRaysPerWave);
    }
    return GNERF_OK;
}

extern "C" int gnerf_render_forward(const gnerf_render_params* p, gnerf_stream_t stream) {
    using namespace gnerf;
    Params P;
    if (int e = fill_params(p, P)) return e;
    if (!p->out_rgb || !p->out_depth || !p->out_wsum || !p->workspace)
        return fail(GNERF_E_ARG, "render: outputs and workspace must not be null");
    const int S = p->depth_resolution, F = p->depth_resolution_importance;
    const int64_t total = P.total_rays;
    hipStream_t s = as_stream(stream);
    // Small launches (one 64x64 frame of gen_videos.py is 256 ray tiles on 256 CUs): a workgroup walking its 16 rays one
    // after the other leaves most of the chip idle and the launch takes 16 ray latencies.  Split each tile over up to four
    // workgroups while that still fits the chip in one wave of workgroups.
    P.split_shift = 0;
    while (P.split_shift < 2 && (int64_t(P.n_tiles) << (P.split_shift + 1)) <= int64_t(kNumCU) * 4) P.split_shift++;
    const int per_xcd = ((P.n_tiles << P.split_shift) + kNumXCD - 1) / kNumXCD;
    const dim3 grid(per_xcd * kNumXCD);
    // Kernel choice (GNERF_RENDER_KERNEL=pipe|coop|generic forces one, for A/B runs):
    //   pipe    3 shader waves + 1 scalar wave, three rays in flight: up to 144+144 samples with importance sampling (1, 2 or 3 tiles per wave)
    //   coop    3 waves per ray, phases separated by barriers: up to 96+96 samples
    //   generic one wave per ray: everything else (up to 256+256)
    // A/B and test overrides, read per call (the parity tests switch kernels inside one process): two getenv walks of the
    // environment, ~0.1 us against the ~10 us of a launch
    const char* force = getenv("GNERF_RENDER_KERNEL");
    const char* force_mlp = getenv("GNERF_RENDER_MLP");                      // f16x3 | f32: overrides params.mlp_mode
    const bool small_planes = int64_t(p->plane_h) * p->plane_w * 3 * 128 < (int64_t(1) << 32);
    bool pipe = P.tiles_c <= 9 && P.tiles_f >= 1 && P.tiles_f <= 9 && small_planes;
    const int pipe_tp = (P.tiles_c <= 3 && P.tiles_f <= 3) ? 1 : ((P.tiles_c <= 6 && P.tiles_f <= 6) ? 2 : 3);   // 16-sample tiles per shader wave and pass
    bool coop = P.tiles_c <= 2 * kCoopWaves && P.tiles_f <= 2 * kCoopWaves && small_planes;
    if (force && !strcmp(force, "generic")) pipe = coop = false;
    if (force && !strcmp(force, "coop")) { pipe = false; if (!coop) return fail(GNERF_E_UNSUPPORTED, "render: cooperative kernel does not cover %d+%d samples", S, F); }
    if (force && !strcmp(force, "pipe") && !pipe) return fail(GNERF_E_UNSUPPORTED, "render: pipelined kernel does not cover %d+%d samples", S, F);
    // Decoder arithmetic of the pipe / coop kernels (the generic kernel is fp32 throughout).
    int mlp = p->mlp_mode;
    if (force_mlp && !strcmp(force_mlp, "f16x3")) mlp = GNERF_MLP_F16X3;
    if (force_mlp && !strcmp(force_mlp, "f32")) mlp = GNERF_MLP_F32;
    if (mlp != GNERF_MLP_AUTO && mlp != GNERF_MLP_F16X3 && mlp != GNERF_MLP_F32) return fail(GNERF_E_ARG, "render: mlp_mode %d is not one of GNERF_MLP_*", mlp);
    if ((pipe || coop) && mlp == GNERF_MLP_AUTO) {
        // the choice is made on the device, by every workgroup for itself (no host round trip, graph-capturable): see choose_mlp
        P.absmax = p->planes_absmax;
        hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(s, &capturing);                   // (the check synchronises the stream: not inside a graph capture)
        if (P.absmax && capturing == hipStreamCaptureStatusNone && getenv("GNERF_VERIFY_ABSMAX") && !strcmp(getenv("GNERF_VERIFY_ABSMAX"), "1")) {
            // debug aid (include/gnerf_hip.h, planes_absmax contract): is the caller's value an upper bound of THESE planes?
            float* own = reinterpret_cast<float*>(static_cast<int*>(p->workspace) + 5);
            if (int e = gnerf_planes_absmax(p->planes_nhwc, int64_t(p->planes_shared ? 1 : p->n_items) * 3 * p->plane_h * p->plane_w * 32, own, stream)) return e;
            float mine = 0.f, theirs = 0.f;
            if (hipMemcpyAsync(&mine, own, 4, hipMemcpyDeviceToHost, s) != hipSuccess || hipMemcpyAsync(&theirs, P.absmax, 4, hipMemcpyDeviceToHost, s) != hipSuccess ||
                hipStreamSynchronize(s) != hipSuccess)
                return fail(GNERF_E_LAUNCH, "render: GNERF_VERIFY_ABSMAX could not read the plane statistics back");
            if (theirs < mine)                                  // (NaN on either side compares false: NaN planes / NaN bound both select fp32)
                return fail(GNERF_E_ARG, "render: planes_absmax = %g is smaller than max |planes| = %g of this call's planes (stale value?)", theirs, mine);
        }
        if (!P.absmax) {
            float* own = reinterpret_cast<float*>(static_cast<int*>(p->workspace) + 5);
            if (int e = gnerf_planes_absmax(p->planes_nhwc, int64_t(p->planes_shared ? 1 : p->n_items) * 3 * p->plane_h * p->plane_w * 32, own, stream)) return e;
            P.absmax = own;
        }
    }
    if (pipe) {
        const int64_t total_seq = P.tiles_per_item > 0 ? int64_t(P.n_tiles) * 16 : total;
        const int per_cu = pipe_tp == 1 ? GNERF_PIPE_WAVES_PER_SIMD : (pipe_tp == 2 ? GNERF_PIPE2_WAVES_PER_SIMD : 2);                         // resident workgroups per CU
        // one dealing unit per workgroup until the chip is full: a small launch is latency-bound, and the three half-steps of
        // pipeline fill cost less than leaving compute units idle (64x64 rays: 167 -> 70 us); the unit shrinks from 8 rays down
        // to what spreads the launch over every resident workgroup slot (a 64x64 frame: 4 rays each on pipe<1>, 6 on pipe<2>)
        const int64_t capacity = int64_t(per_cu) * kNumCU;
        P.pipe_unit = kPipeUnit;
        if (total_seq < capacity * kPipeUnit) P.pipe_unit = int((total_seq + capacity - 1) / capacity);       // >= 1: total_seq >= 1
        int64_t g = ((total_seq + P.pipe_unit - 1) / P.pipe_unit + kNumXCD - 1) / kNumXCD * kNumXCD;
        if (g < kNumXCD) g = kNumXCD;
        if (g > capacity) g = capacity;
        const bool gen = !p->ray_origins || p->rng_mode != GNERF_RNG_TENSORS;       // the call makes its rays and / or its draws in the kernel
        const size_t lds_bytes = pipe_lds_floats(pipe_tp, mlp, gen) * sizeof(float);
        const dim3 gd((unsigned)g), bd(kPipeThreads);
        // the instantiation with compile-time sample counts (render_pipe_body<.., FULL>) where the call fills the slots exactly
        bool full = S == 48 * pipe_tp && F == 48 * pipe_tp && !p->disparity_space_sampling && !p->ray_start_per_ray && !p->debug && !p->sigma_noise_coarse;
#ifdef GNERF_STAMPS
        full = full || (S == 48 * pipe_tp && F == 48 * pipe_tp && !p->disparity_space_sampling && !p->ray_start_per_ray);     // the timing build's `debug` is its stamp buffer
#endif
        if (const char* f = getenv("GNERF_PIPE_FULL")) full = full && strcmp(f, "0") != 0;       // A/B runs and the tests' cross-check
        if (gen) {
            if (!full || pipe_tp > 2)
                return fail(GNERF_E_UNSUPPORTED, "render: in-kernel rays / draws are built for 48+48 and 96+96 samples with plain stratified sampling (got %d+%d)", S, F);
#define GNERF_PIPE_GEN(TP) do { if (mlp == kMlpAuto) hipLaunchKernelGGL((render_kernel_pipe<TP, kMlpAuto, true, true>), gd, bd, lds_bytes, s, P); \
                                else if (mlp == kMlpF16x3) hipLaunchKernelGGL((render_kernel_pipe<TP, kMlpF16x3, true, true>), gd, bd, lds_bytes, s, P); \
                                else hipLaunchKernelGGL((render_kernel_pipe<TP, kMlpF32, true, true>), gd, bd, lds_bytes, s, P); } while (0)
            if (pipe_tp == 1) GNERF_PIPE_GEN(1); else GNERF_PIPE_GEN(2);
#undef GNERF_PIPE_GEN
        } else {
#define GNERF_PIPE2(TP, FULL) do { if (mlp == kMlpAuto) hipLaunchKernelGGL((render_kernel_pipe<TP, kMlpAuto, FULL>), gd, bd, lds_bytes, s, P); \
                            else if (mlp == kMlpF16x3) hipLaunchKernelGGL((render_kernel_pipe<TP, kMlpF16x3, FULL>), gd, bd, lds_bytes, s, P); \
                            else hipLaunchKernelGGL((render_kernel_pipe<TP, kMlpF32, FULL>), gd, bd, lds_bytes, s, P); } while (0)
#define GNERF_PIPE(TP) do { if (full) GNERF_PIPE2(TP, true); else GNERF_PIPE2(TP, false); } while (0)
        if (pipe_tp == 1) GNERF_PIPE(1);
        else if (pipe_tp == 2) GNERF_PIPE(2);
        else {                                              // 65 KB of LDS: above the default dynamic limit
            static PerDeviceOnce once3[6];
            if (mlp == kMlpAuto) { if (int e = full ? once3[0].raise_lds(render_kernel_pipe<3, kMlpAuto, true>, "render") : once3[3].raise_lds(render_kernel_pipe<3, kMlpAuto, false>, "render")) return e; }
            else if (mlp == kMlpF16x3) { if (int e = full ? once3[1].raise_lds(render_kernel_pipe<3, kMlpF16x3, true>, "render") : once3[4].raise_lds(render_kernel_pipe<3, kMlpF16x3, false>, "render")) return e; }
            else { if (int e = full ? once3[2].raise_lds(render_kernel_pipe<3, kMlpF32, true>, "render") : once3[5].raise_lds(render_kernel_pipe<3, kMlpF32, false>, "render")) return e; }
            GNERF_PIPE(3);
        }
#undef GNERF_PIPE
#undef GNERF_PIPE2
        }
        if (int e = check_launch("render_kernel_pipe")) return e;
    } else
    if (!p->ray_origins || p->rng_mode != GNERF_RNG_TENSORS) {
        return fail(GNERF_E_UNSUPPORTED, "render: in-kernel rays / draws need the pipelined kernel (48+48 or 96+96 samples); got %d+%d", S, F);
    } else
    if (coop) {
        const int tc1 = (P.tiles_c + kCoopWaves - 1) / kCoopWaves, tf1 = (P.tiles_f + kCoopWaves - 1) / kCoopWaves;
        const dim3 block(kCoopThreads);
        const size_t lds_bytes = coop_lds_floats(16 * (P.tiles_c + P.tiles_f), mlp) * sizeof(float);
#define GNERF_COOP(TC, TF) do { if (mlp == kMlpAuto) hipLaunchKernelGGL((render_kernel_coop<TC, TF, kMlpAuto>), grid, block, lds_bytes, s, P); \
                                else if (mlp == kMlpF16x3) hipLaunchKernelGGL((render_kernel_coop<TC, TF, kMlpF16x3>), grid, block, lds_bytes, s, P); \
                                else hipLaunchKernelGGL((render_kernel_coop<TC, TF, kMlpF32>), grid, block, lds_bytes, s, P); } while (0)
        if (tc1 == 1 && tf1 == 0) GNERF_COOP(1, 0);
        else if (tc1 == 1 && tf1 == 1) GNERF_COOP(1, 1);
        else if (tc1 == 1 && tf1 == 2) GNERF_COOP(1, 2);
        else if (tc1 == 2 && tf1 == 0) GNERF_COOP(2, 0);
        else if (tc1 == 2 && tf1 == 1) GNERF_COOP(2, 1);
        else GNERF_COOP(2, 2);
#undef GNERF_COOP
        if (int e = check_launch("render_kernel_coop")) return e;
    } else {
        const size_t lds_bytes = scratch_floats(16 * (P.tiles_c + P.tiles_f), P.tiles_c + P.tiles_f) * sizeof(float);
        if (lds_bytes > 160 * 1024) return fail(GNERF_E_ARG, "render: %d+%d samples need %zu bytes of LDS (> 160 KiB)", S, F, lds_bytes);
        static PerDeviceOnce once;
        if (int e = once.raise_lds(render_kernel_generic, "render")) return e;
        hipLaunchKernelGGL(render_kernel_generic, grid, dim3(64), lds_bytes, s, P);
        if (int e = check_launch("render_kernel_generic")) return e;
    }
    hipLaunchKernelGGL(clamp_depth_kernel, dim3((unsigned)((total + 256 * kClampPerThread - 1) / (256 * kClampPerThread))), dim3(256), 0, s,
                       p->out_depth, static_cast<unsigned*>(p->workspace), total, p->rays_per_item, p->depth_clamp_per_item ? p->n_items : 0);
    return check_launch("clamp_depth_kernel");
}

extern "C" int gnerf_render_backward(const gnerf_render_params* p, const gnerf_render_grads* g, gnerf_stream_t stream) {
    using namespace gnerf;
    Params P;
    if (int e = fill_params(p, P)) return e;
    if (!g) return fail(GNERF_E_ARG, "render_backward: grads is null");
    if (p->planes_shared) return fail(GNERF_E_UNSUPPORTED, "render_backward: planes_shared is a forward-only option");
    if (!p->ray_origins || p->rng_mode != GNERF_RNG_TENSORS) return fail(GNERF_E_UNSUPPORTED, "render_backward: in-kernel rays / draws are forward-only options");
    if (p->sigma_noise_coarse || p->sigma_noise_fine) return fail(GNERF_E_UNSUPPORTED, "render_backward: density noise is a forward-only option");
    const int n_dec = (g->grad_w1 != nullptr) + (g->grad_b1 != nullptr) + (g->grad_w2 != nullptr) + (g->grad_b2 != nullptr);
    if (n_dec != 0 && n_dec != 4) return fail(GNERF_E_ARG, "render_backward: the four decoder gradients are given together or not at all");
    if (!g->grad_planes_nhwc && n_dec == 0) return GNERF_OK;
    if (!(int64_t(p->plane_h) * p->plane_w * 3 * 128 < (int64_t(1) << 32))) return fail(GNERF_E_UNSUPPORTED, "render_backward: planes too large for 32-bit tap offsets");
    static_assert(kBwdRaysPerWave == kRaysPerWave, "ray tiles are shared with the forward launcher");
    const size_t lds_bytes = (kBwdWeightFloats + kBwdWaves * bwd_wave_floats(16 * (P.tiles_c + P.tiles_f))) * sizeof(float);
    if (lds_bytes > 160 * 1024) return fail(GNERF_E_ARG, "render_backward: %d+%d samples need %zu bytes of LDS (> 160 KiB)", p->depth_resolution, p->depth_resolution_importance, lds_bytes);
    static PerDeviceOnce once;
    static PerDeviceOnce once_staged;
    if (int e = once.raise_lds(render_bwd_kernel<false>, "render_backward")) return e;
    if (int e = once_staged.raise_lds(render_bwd_kernel<true>, "render_backward")) return e;
    P.split_shift = 0;                                       // up to two workgroups of four waves per CU: share tiles until the chip is full
    while (P.split_shift < 2 && (int64_t(P.n_tiles) << (P.split_shift + 1)) <= int64_t(kNumCU) * 2 * kBwdWaves) P.split_shift++;
    const int n_blocks = ((P.n_tiles << P.split_shift) + kBwdWaves - 1) / kBwdWaves;
    const int per_xcd = (n_blocks + kNumXCD - 1) / kNumXCD;
    // Staged scatter (see plane_scatter_kernel): needs the caller's staging buffer, a plane gradient to make, and ray tiles that do
    // not straddle items.  GNERF_BWD_SCATTER=direct|staged forces one route (A/B runs and tests).
    const char* route = getenv("GNERF_BWD_SCATTER");
    bool tiles_ok = P.tiles_per_item > 0 || p->n_items == 1 || p->rays_per_item % kBwdRaysPerWave == 0;
    {
        // Round 6: a ragged call (several items whose ray count is no multiple of 16) no longer drops to the one-wave-per-ray kernel with
        // float atomics: where the pipelined kernels and the binned scatter cover the shape, the ray SEQUENCE pads every item to whole 16-ray
        // tiles (pipe_seq_to_ray / bin_tile_ray answer -1 for the padding, which every kernel of the path already skips)
        const char* bk = getenv("GNERF_BWD_KERNEL");
        const int n_all_r = p->depth_resolution + p->depth_resolution_importance;
        const bool piped_shape = P.tiles_c <= 9 && P.tiles_f <= 9 && !(bk && !strcmp(bk, "wave")) && !(route && (!strcmp(route, "direct") || !strcmp(route, "sorted")));
        const bool binned_shape = int64_t(P.total_rays) * n_all_r * 33 < (int64_t(1) << 32) &&
                                  (2 * size_t(3) * ((p->plane_h + kBinTile - 1) / kBinTile) * ((p->plane_w + kBinTile - 1) / kBinTile) + 128) * 4 <= 150 * 1024;
        const int pad = (p->rays_per_item + kBwdRaysPerWave - 1) / kBwdRaysPerWave * kBwdRaysPerWave;
        if (!tiles_ok && g->scatter_stage && g->grad_planes_nhwc && piped_shape && binned_shape && int64_t(p->n_items) * pad < INT32_MAX) {
            P.tiles_y = pad;                                        // (tiles_per_item == 0: linear_pad(P))
            P.n_tiles = p->n_items * (pad / kBwdRaysPerWave);
            tiles_ok = true;
        }
    }
    bool staged = g->scatter_stage != nullptr && g->grad_planes_nhwc != nullptr && tiles_ok;
    if (route && !strcmp(route, "direct")) staged = false;
    if (route && !strcmp(route, "staged") && !staged) return fail(GNERF_E_ARG, "render_backward: the staged scatter needs scatter_stage, a plane gradient and whole tiles per item");
    // Pipelined path of the staged form (round 4): the ray-level part on the forward pipeline (render_kernel_pipe_bwd), the per-sample
    // part as a kernel over sample tiles (render_bwd_tiles_kernel).  Shapes the pipelined kernels cover; GNERF_BWD_KERNEL=wave keeps the
    // one-wave-per-ray kernel (A/B runs and the tests' cross-check).
    const char* bwd_kernel = getenv("GNERF_BWD_KERNEL");
    const bool small_planes = int64_t(p->plane_h) * p->plane_w * 3 * 128 < (int64_t(1) << 32);
    // (a decoder-only request with an exchange buffer -- gnerf_render_backward_exchange_bytes -- takes the same two kernels: no dX rows,
    // no second pass)
    const bool exchange_only = g->scatter_stage != nullptr && g->grad_planes_nhwc == nullptr && n_dec == 4;
    // (round 6: F = 0 too -- the pipeline's steps and barriers do not depend on the sample counts; a ray without an importance pass shades no
    //  fine tile, merges nothing and marches its coarse samples)
    const bool piped = (staged || exchange_only) && P.tiles_c <= 9 && P.tiles_f <= 9 && small_planes && !(bwd_kernel && !strcmp(bwd_kernel, "wave"))
                       && !(route && !strcmp(route, "direct"));
    if (piped) {
        hipStream_t s = as_stream(stream);
        const int64_t total = P.total_rays;
        const int n_all = p->depth_resolution + p->depth_resolution_importance;
        // decoder arithmetic of the first pass: chosen on the device from max |planes| like the forward's (measured here into the
        // 256 bytes behind the staged rows when the caller has none)
        const int tiles_per_ray = (n_all + 15) / 16;
        P.bwd_ray_stride = staged ? int64_t(n_all) * 33 : int64_t(n_all) + 32 * tiles_per_ray;
        P.bwd_tile_pitch = staged ? 512 : 32;
        P.absmax = p->planes_absmax;
        if (!P.absmax) {
            float* own = g->scatter_stage + size_t(total) * P.bwd_ray_stride;
            if (int e = gnerf_planes_absmax(p->planes_nhwc, int64_t(p->n_items) * 3 * p->plane_h * p->plane_w * 32, own, stream)) return e;
            P.absmax = own;
        }
        P.p.workspace = nullptr;                                   // (the backward's params carry no workspace)
        // Decoder arithmetic: both kernels choose on the device like the forward (f16 hi/lo products when features, weights and
        // activations are in f16's range, exact fp32 otherwise).  GNERF_BWD_MLP=f32|f16x3|auto forces both, _K1 / _K2 one of them.
        P.p.mlp_mode = GNERF_MLP_AUTO;
        Params P2 = P;
        auto mode_of = [](const char* v, int dflt) { return !v ? dflt : !strcmp(v, "f32") ? GNERF_MLP_F32 : !strcmp(v, "f16x3") ? GNERF_MLP_F16X3 : !strcmp(v, "auto") ? GNERF_MLP_AUTO : dflt; };
        P.p.mlp_mode = mode_of(getenv("GNERF_BWD_MLP"), P.p.mlp_mode);
        P2.p.mlp_mode = mode_of(getenv("GNERF_BWD_MLP"), P2.p.mlp_mode);
        P.p.mlp_mode = mode_of(getenv("GNERF_BWD_MLP_K1"), P.p.mlp_mode);
        P2.p.mlp_mode = mode_of(getenv("GNERF_BWD_MLP_K2"), P2.p.mlp_mode);
        const int pipe_tp = (P.tiles_c <= 3 && P.tiles_f <= 3) ? 1 : ((P.tiles_c <= 6 && P.tiles_f <= 6) ? 2 : 3);
        const int64_t total_seq = (P.tiles_per_item > 0 || linear_pad(P) > 0) ? int64_t(P.n_tiles) * 16 : total;
        const int per_cu = pipe_tp == 1 ? GNERF_PIPE_WAVES_PER_SIMD : (pipe_tp == 2 ? GNERF_PIPE2_WAVES_PER_SIMD : 2);
        const int64_t capacity = int64_t(per_cu) * kNumCU;
        P.pipe_unit = kPipeUnit;
        if (total_seq < capacity * kPipeUnit) P.pipe_unit = int((total_seq + capacity - 1) / capacity);
        int64_t gsz = ((total_seq + P.pipe_unit - 1) / P.pipe_unit + kNumXCD - 1) / kNumXCD * kNumXCD;
        if (gsz < kNumXCD) gsz = kNumXCD;
        if (gsz > capacity) gsz = capacity;
        const size_t lds1 = pipe_lds_floats(pipe_tp, kMlpAuto) * sizeof(float);
        if (pipe_tp == 1) hipLaunchKernelGGL(render_kernel_pipe_bwd<1>, dim3((unsigned)gsz), dim3(kPipeThreads), lds1, s, P, *g, g->scatter_stage);
        else if (pipe_tp == 2) hipLaunchKernelGGL(render_kernel_pipe_bwd<2>, dim3((unsigned)gsz), dim3(kPipeThreads), lds1, s, P, *g, g->scatter_stage);
        else {
            static PerDeviceOnce once3;
            if (int e = once3.raise_lds(render_kernel_pipe_bwd<3>, "render_backward")) return e;
            hipLaunchKernelGGL(render_kernel_pipe_bwd<3>, dim3((unsigned)gsz), dim3(kPipeThreads), lds1, s, P, *g, g->scatter_stage);
        }
        if (int e = check_launch("render_kernel_pipe_bwd")) return e;
        const size_t lds2 = (bwd_tiles_weight_floats() + kTileWaves * bwd_tiles_wave_floats()) * sizeof(float);
        static PerDeviceOnce once_tiles;
        if (int e = once_tiles.raise_lds(render_bwd_tiles_kernel, "render_backward")) return e;
        const int64_t sample_tiles = total_seq * ((n_all + 15) / 16);
        int64_t g2 = (sample_tiles + kTileWaves - 1) / kTileWaves;
        if (g2 > int64_t(kNumCU) * 2) g2 = int64_t(kNumCU) * 2;    // two workgroups of four waves per CU, each walking a contiguous run of tiles
        g2 = (g2 + kNumXCD - 1) / kNumXCD * kNumXCD;
        P2.pipe_unit = P.pipe_unit;
        hipLaunchKernelGGL(render_bwd_tiles_kernel, dim3((unsigned)g2), dim3(kTileThreads), lds2, s, P2, *g, g->scatter_stage);
        if (int e = check_launch("render_bwd_tiles_kernel")) return e;
    } else {
    if (staged) hipLaunchKernelGGL(render_bwd_kernel<true>, dim3(per_xcd * kNumXCD), dim3(kBwdThreads), lds_bytes, as_stream(stream), P, *g, g->scatter_stage);
    else        hipLaunchKernelGGL(render_bwd_kernel<false>, dim3(per_xcd * kNumXCD), dim3(kBwdThreads), lds_bytes, as_stream(stream), P, *g, static_cast<float*>(nullptr));
    if (int e = check_launch("render_bwd_kernel")) return e;
    }
    // Second pass of the staged form.  Round 5: bin the rows by plane tile and sum each tile in LDS (scatter_binned.inl: no global float
    // atomics, bit-reproducible); GNERF_BWD_SCATTER=sorted keeps round 2's per-ray-tile sort with one atomic per texel and chunk
    // (plane_scatter_kernel), which also takes the calls the binned form does not cover.
    if (staged) {
        const int n_all_s = p->depth_resolution + p->depth_resolution_importance;
        P.bwd_ray_stride = int64_t(n_all_s) * 33;
        const bool binned_ok = int64_t(P.total_rays) * n_all_s * 33 < (int64_t(1) << 32) &&
                               (2 * size_t(3) * ((p->plane_h + kBinTile - 1) / kBinTile) * ((p->plane_w + kBinTile - 1) / kBinTile) + 128) * 4 <= 150 * 1024;
        if (binned_ok && !(route && !strcmp(route, "sorted"))) {
            char* ws = reinterpret_cast<char*>(g->scatter_stage) + bin_workspace_offset(P.total_rays, n_all_s);
            return launch_binned_scatter(P, g->scatter_stage, ws, g->grad_planes_nhwc, as_stream(stream));
        }
        static PerDeviceOnce once_scatter;
        if (int e = once_scatter.raise_lds(plane_scatter_kernel, "render_backward")) return e;
        hipLaunchKernelGGL(plane_scatter_kernel, dim3(P.n_tiles), dim3(kScatterThreads), scatter_lds_floats() * sizeof(float), as_stream(stream),
                           P, static_cast<const float*>(g->scatter_stage), g->grad_planes_nhwc);
        return check_launch("plane_scatter_kernel");
    }
    return GNERF_OK;
}

extern "C" size_t gnerf_render_backward_stage_bytes(const gnerf_render_params* p) {
    if (!p || p->n_items < 1 || p->rays_per_item < 1) return 0;
    const size_t n_all = size_t(p->depth_resolution) + size_t(p->depth_resolution_importance);
    const int64_t rays = int64_t(p->n_items) * p->rays_per_item;
    // the staged rows (+ a spare line: max |planes| when the caller has none), then the binned scatter's workspace: counters, the
    // record array (16 bytes per sample and plane) and the tiles' halos (scatter_binned.inl)
    return bin_workspace_offset(rays, int(n_all)) + bin_workspace_bytes(rays, int(n_all), p->n_items, p->plane_h > 0 ? p->plane_h : 1, p->plane_w > 0 ? p->plane_w : 1);
}

extern "C" size_t gnerf_render_backward_exchange_bytes(const gnerf_render_params* p) {
    if (!p || p->n_items < 1 || p->rays_per_item < 1) return 0;
    const size_t n_all = size_t(p->depth_resolution) + size_t(p->depth_resolution_importance);
    return size_t(p->n_items) * size_t(p->rays_per_item) * (n_all + 32 * ((n_all + 15) / 16)) * sizeof(float) + 256;
}

extern "C" int gnerf_query_points(const float* planes_nhwc, int n_items, int plane_h, int plane_w,
                                  const float* points, int n_points, float box_warp,
                                  const float* w1, const float* b1, const float* w2, const float* b2,
                                  float* out_sigma, float* out_rgb, int planes_interleaved, gnerf_stream_t stream) {
    using namespace gnerf;
    gnerf_render_params p = {};
    p.planes_interleaved = planes_interleaved;
    p.planes_nhwc = planes_nhwc; p.n_items = n_items; p.plane_h = plane_h; p.plane_w = plane_w;
    p.w1 = w1; p.b1 = b1; p.w2 = w2; p.b2 = b2; p.box_warp = box_warp;
    if (int e = check_common(&p)) return e;
    if (!points || !out_sigma) return fail(GNERF_E_ARG, "query_points: null pointer");
    if (n_points < 1) return fail(GNERF_E_ARG, "query_points: n_points must be positive");
    const int64_t tiles = int64_t(n_items) * ((n_points + 15) / 16);
    if (tiles > INT32_MAX) return fail(GNERF_E_ARG, "query_points: too many points");
    const int blocks = int(tiles < int64_t(kNumCU) * 16 ? tiles : int64_t(kNumCU) * 16);
    hipLaunchKernelGGL(query_kernel, dim3(blocks), dim3(64), 0, as_stream(stream), p, float(2.0 / double(box_warp)), n_points, int(tiles),
                       points, out_sigma, out_rgb);
    return check_launch("query_kernel");
}

extern "C" int gnerf_query_points_backward(const float* planes_nhwc, int n_items, int plane_h, int plane_w,
                                           const float* points, int n_points, float box_warp,
                                           const float* w1, const float* b1, const float* w2, const float* b2,
                                           const float* grad_sigma, const float* grad_rgb,
                                           float* grad_planes_nhwc, float* grad_w1, float* grad_b1, float* grad_w2, float* grad_b2,
                                           int planes_interleaved, gnerf_stream_t stream) {
    using namespace gnerf;
    Params P = {};
    gnerf_render_params& p = P.p;
    p.planes_interleaved = planes_interleaved;
    p.planes_nhwc = planes_nhwc; p.n_items = n_items; p.plane_h = plane_h; p.plane_w = plane_w;
    p.w1 = w1; p.b1 = b1; p.w2 = w2; p.b2 = b2; p.box_warp = box_warp;
    if (int e = check_common(&p)) return e;
    if (!points) return fail(GNERF_E_ARG, "query_points_backward: points is null");
    if (n_points < 1) return fail(GNERF_E_ARG, "query_points_backward: n_points must be positive");
    const int n_dec = (grad_w1 != nullptr) + (grad_b1 != nullptr) + (grad_w2 != nullptr) + (grad_b2 != nullptr);
    if (n_dec != 0 && n_dec != 4) return fail(GNERF_E_ARG, "query_points_backward: the four decoder gradients are given together or not at all");
    if ((!grad_planes_nhwc && n_dec == 0) || (!grad_sigma && !grad_rgb)) return GNERF_OK;
    if (!(int64_t(plane_h) * plane_w * 3 * 128 < (int64_t(1) << 32))) return fail(GNERF_E_UNSUPPORTED, "query_points_backward: planes too large for 32-bit tap offsets");
    P.box_scale = float(2.0 / double(box_warp));
    fill_pitches(P);
    QueryBwdArgs Q;
    Q.points = points; Q.grad_sigma = grad_sigma; Q.grad_rgb = grad_rgb; Q.n_points = n_points;
    Q.tiles_per_item = (n_points + 15) / 16;
    const int64_t tiles = int64_t(n_items) * Q.tiles_per_item;
    if (tiles > INT32_MAX) return fail(GNERF_E_ARG, "query_points_backward: too many points");
    Q.n_tiles = int(tiles);
    Q.grad_planes_nhwc = grad_planes_nhwc; Q.grad_w1 = grad_w1; Q.grad_b1 = grad_b1; Q.grad_w2 = grad_w2; Q.grad_b2 = grad_b2;
    const size_t lds_bytes = (kBwdWeightFloats + kBwdWaves * bwd_wave_floats(0)) * sizeof(float);
    int64_t blocks = (tiles + kBwdWaves - 1) / kBwdWaves;
    if (blocks > int64_t(kNumCU) * 2) blocks = int64_t(kNumCU) * 2;
    hipLaunchKernelGGL(query_bwd_kernel, dim3((unsigned)blocks), dim3(kBwdThreads), lds_bytes, as_stream(stream), P, Q);
    return check_launch("query_bwd_kernel");
}
