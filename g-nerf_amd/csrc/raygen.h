// Device functions shared by the kernels that make rays and uniform draws themselves (csrc/planes.hip: gnerf_make_rays,
// gnerf_torch_rand; csrc/render_pipe.inl: the in-kernel forms, gnerf_render_params.cam2world / rng_mode).
#pragma once

#include "common.h"

namespace gnerf {

// RaySampler.forward (training/volumetric_rendering/ray_sampler.py:24-63) for the ray through pixel (row, col) of a res x res image:
// direction = normalise(cam2world @ (x_lift, y_lift, 1, 1) - cam_loc); the origin is cam_loc = M[:, 3].  Arithmetic order follows
// ray_sampler.py:43-59 with every operation rounded on its own (no fused multiply-adds), so that the directions agree with the
// reference to the last bit or two and every caller of this function gets the same bits.
__device__ __forceinline__ void camera_ray(const float* __restrict__ M, const float* __restrict__ K, int res, int row, int col, float (&dir)[3]) {
    const float fx = K[0], sk = K[1], cx = K[2], fy = K[4], cy = K[5];
    const float inv = 1.0f / float(res), half = 0.5f / float(res);
    const float xc = __fadd_rn(__fmul_rn(float(col), inv), half);
    const float yc = __fadd_rn(__fmul_rn(float(row), inv), half);
    // x_lift = (x - cx + cy*sk/fy - sk*y/fy) / fx ;  y_lift = (y - cy) / fy          ray_sampler.py:51-52
    float xl = __fsub_rn(xc, cx);
    xl = __fadd_rn(xl, __fdiv_rn(__fmul_rn(cy, sk), fy));
    xl = __fsub_rn(xl, __fdiv_rn(__fmul_rn(sk, yc), fy));
    xl = __fdiv_rn(xl, fx);
    const float yl = __fdiv_rn(__fsub_rn(yc, cy), fy);
    float w[3];
#pragma unroll
    for (int r = 0; r < 3; r++) {
        // row r of cam2world times (xl, yl, 1, 1)
        float acc = __fmul_rn(M[r * 4 + 0], xl);
        acc = __fadd_rn(acc, __fmul_rn(M[r * 4 + 1], yl));
        acc = __fadd_rn(acc, M[r * 4 + 2]);
        acc = __fadd_rn(acc, M[r * 4 + 3]);
        w[r] = __fsub_rn(acc, M[r * 4 + 3]);
    }
    float nrm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(w[0], w[0]), __fmul_rn(w[1], w[1])), __fmul_rn(w[2], w[2])));
    nrm = fmaxf(nrm, 1e-12f);       // F.normalize eps
#pragma unroll
    for (int r = 0; r < 3; r++) dir[r] = __fdiv_rn(w[r], nrm);
}

// Philox4x32-10 (Salmon et al., SC'11; what rocRAND's philox4x32_10 engine and cuRAND's compute): ten rounds on the 128-bit counter
// (c0 lowest word) with the 64-bit key (k0, k1); returns output word `pick` (0..3) -- the compiler drops the last round's unused half.
__device__ __forceinline__ uint32_t philox4x32_10_word(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t pick) {
#pragma unroll
    for (int round = 0; round < 10; round++) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return pick == 0 ? c0 : (pick == 1 ? c1 : (pick == 2 ? c2 : c3));
}

// The whole block (the four output words of one counter value): what one thread of ATen's kernel computes per call
__device__ __forceinline__ void philox4x32_10_block(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t (&out)[4]) {
#pragma unroll
    for (int round = 0; round < 10; round++) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// One draw of `torch.rand(numel, device)` with the device generator at (seed, philox offset): launch-uniform description.
// ctr = offset / 4 (the generator hands out offsets in multiples of four 32-bit words); threads = ATen's grid size in threads,
// either a power of two (log2 in `shift`) or >= numel (`shift` = 63: every element is its own thread's first word).
struct TorchRandDraw {
    uint32_t k0, k1;        // seed
    uint64_t ctr;           // offset / 4
    uint32_t mask;          // threads - 1 (power-of-two case) or 0xffffffff
    uint32_t shift;         // log2(threads), or 63
};

// Element `li` of the draw: ATen's distribution_elementwise_grid_stride_kernel (unroll 4) gives element li to thread li % threads,
// call (li / threads) / 4, word (li / threads) % 4 of that call's block; rocRAND's float uniform is x * 2^-32 + 2^-32 in (0, 1] and
// ATen's uniform_ maps 1 back to 0 (DistributionTemplates.h, "reverse the bounds").  Restated and pinned in oracle/philox_ref.py.
__device__ __forceinline__ float torch_rand_element(const TorchRandDraw& d, uint64_t li) {
    const uint32_t tid = uint32_t(li) & d.mask;
    const uint32_t m = uint32_t(li >> d.shift);
    const uint64_t c = d.ctr + (m >> 2);
    const uint32_t x = philox4x32_10_word(uint32_t(c), uint32_t(c >> 32), tid, 0u, d.k0, d.k1, m & 3u);
    const float u = __fmaf_rn(float(x), 0x1p-32f, 0x1p-32f);
    return u == 1.0f ? 0.0f : u;
}

// Host side: the description of a draw (false when ATen's geometry for it is neither of the two supported cases).
inline bool torch_rand_draw(uint64_t seed, uint64_t offset, uint32_t threads, int64_t numel, TorchRandDraw& d) {
    d.k0 = uint32_t(seed); d.k1 = uint32_t(seed >> 32);
    d.ctr = offset / 4;
    if (offset % 4 != 0 || threads == 0) return false;
    if (int64_t(threads) >= numel) { d.mask = 0xffffffffu; d.shift = 63; return numel <= int64_t(0xffffffffu); }
    if (threads & (threads - 1)) return false;
    d.mask = threads - 1;
    d.shift = 0;
    while ((uint32_t(1) << d.shift) < threads) d.shift++;
    return true;
}

}  // namespace gnerf
