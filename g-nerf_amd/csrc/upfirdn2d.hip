// upfirdn2d for gfx950: zero-insert upsample -> pad/crop -> 2-D FIR -> decimate.
//
// Replaces upfirdn2d_plugin.upfirdn2d (reference torch_utils/ops/upfirdn2d.cpp:20, upfirdn2d.cu:33-379);
// semantics follow torch_utils/ops/upfirdn2d.py:168-213:
//     y[oy,ox] = gain * sum_{ky,kx} k[ky,kx] * z[oy*downy + ky, ox*downx + kx]
// where z is x zero-upsampled by (upx,upy) and padded by (padx0,pady0) on the low side (negative =
// crop), and k is f flipped in both axes unless `flip` (flip=0 is a true convolution).
//
// Two kernels:
//   * upfirdn_tile_kernel  -- NCHW-contiguous images and filters up to 8x8 (G-NeRF only ever uses the
//     4x4 [1,3,3,1] filter): a workgroup owns a TILE_W x TILE_H output tile of one (n,c) image, stages
//     the input window it needs in LDS with coalesced row reads ("line buffer"), keeps the filter in
//     LDS too, and each lane produces OUT_PER_LANE horizontally adjacent outputs so that LDS reads are
//     shared between them.  HBM traffic is the algorithmic in+out plus the halo.
//   * upfirdn_any_kernel   -- any strides (channels_last), any filter size, straight from global memory
//     (L1/L2 absorb the tap re-reads).

#include "common.h"
#include <cstdlib>

namespace {

using namespace gnerf;

template <class T> struct alignas(16) Pack16 { T v[16 / sizeof(T)]; };

struct UpArgs {
    const void* x; const float* f; void* y;
    int n, c, in_h, in_w;
    int64_t xs_n, xs_c, xs_h, xs_w;
    int fh, fw; int64_t fs_h, fs_w;
    int out_h, out_w;
    int64_t ys_n, ys_c, ys_h, ys_w;
    int upx, upy, downx, downy, padx0, pady0, flip;
    float gain;
};

__device__ __forceinline__ int floor_div_pos(int a, int b) {   // floor(a / b) for b > 0, any a
    int q = a / b;
    return (a % b < 0) ? q - 1 : q;
}
__device__ __forceinline__ int ceil_div_pos(int a, int b) { return -floor_div_pos(-a, b); }

// ---------------------------------------------------------------------------------------------
// Generic kernel: one output element per lane, lanes ordered like y's memory.

template <class T>
__global__ __launch_bounds__(256) void upfirdn_any_kernel(UpArgs a, int channels_last) {
    typedef typename Arith<T>::type A;
    const T* x = static_cast<const T*>(a.x);
    T* y = static_cast<T*>(a.y);
    const int64_t total = int64_t(a.n) * a.c * a.out_h * a.out_w;
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += stride) {
        int ox, oy, ch, img;
        int64_t r = i;
        if (channels_last) { ch = int(r % a.c); r /= a.c; ox = int(r % a.out_w); r /= a.out_w; oy = int(r % a.out_h); img = int(r / a.out_h); }
        else               { ox = int(r % a.out_w); r /= a.out_w; oy = int(r % a.out_h); r /= a.out_h; ch = int(r % a.c); img = int(r / a.c); }
        // taps ky with (oy*downy + ky - pady0) = iy*upy, 0 <= iy < in_h
        const int by = oy * a.downy - a.pady0, bx = ox * a.downx - a.padx0;
        int iy_lo = ceil_div_pos(by, a.upy);            if (iy_lo < 0) iy_lo = 0;
        int iy_hi = floor_div_pos(by + a.fh - 1, a.upy); if (iy_hi > a.in_h - 1) iy_hi = a.in_h - 1;
        int ix_lo = ceil_div_pos(bx, a.upx);            if (ix_lo < 0) ix_lo = 0;
        int ix_hi = floor_div_pos(bx + a.fw - 1, a.upx); if (ix_hi > a.in_w - 1) ix_hi = a.in_w - 1;
        const T* xp = x + img * a.xs_n + ch * a.xs_c;
        A acc = A(0);
        for (int iy = iy_lo; iy <= iy_hi; iy++) {
            const int ky = iy * a.upy - by;
            const int fy = a.flip ? ky : a.fh - 1 - ky;
            for (int ix = ix_lo; ix <= ix_hi; ix++) {
                const int kx = ix * a.upx - bx;
                const int fx = a.flip ? kx : a.fw - 1 - kx;
                acc += load_as<T>(xp, iy * a.xs_h + ix * a.xs_w) * A(a.f[fy * a.fs_h + fx * a.fs_w]);
            }
        }
        store_as<T>(y, img * a.ys_n + ch * a.ys_c + oy * a.ys_h + ox * a.ys_w, acc * A(a.gain));
    }
}

// ---------------------------------------------------------------------------------------------
// Tiled kernel for contiguous NCHW.

constexpr int TILE_W = 64, TILE_H = 16, OUT_PER_LANE = 4;    // 256 lanes: 16 column groups x 16 rows
constexpr int MAX_TAPS = 8;
// worst-case input window: ceil((TILE-1)*down + taps) / up) + 1, bounded here for down <= 2, up >= 1
constexpr int WIN_W = (TILE_W - 1) * 2 + MAX_TAPS + 1, WIN_H = (TILE_H - 1) * 2 + MAX_TAPS + 1;

template <class T>
__global__ __launch_bounds__(256) void upfirdn_tile_kernel(UpArgs a, int tiles_x, int tiles_y) {
    typedef typename Arith<T>::type A;
    __shared__ float s_f[MAX_TAPS * MAX_TAPS];
    __shared__ float s_x[WIN_H * (WIN_W + 1)];
    const T* x = static_cast<const T*>(a.x);
    T* y = static_cast<T*>(a.y);
    int t = blockIdx.x;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y; t /= tiles_y;
    const int64_t img = t;                                     // n*c + ch
    const int ox0 = tx * TILE_W, oy0 = ty * TILE_H;
    const int tw = min(TILE_W, a.out_w - ox0), th = min(TILE_H, a.out_h - oy0);
    // input window needed by this output tile
    const int by0 = oy0 * a.downy - a.pady0, bx0 = ox0 * a.downx - a.padx0;
    int iy_lo = max(ceil_div_pos(by0, a.upy), 0);
    int iy_hi = min(floor_div_pos(by0 + (th - 1) * a.downy + a.fh - 1, a.upy), a.in_h - 1);
    int ix_lo = max(ceil_div_pos(bx0, a.upx), 0);
    int ix_hi = min(floor_div_pos(bx0 + (tw - 1) * a.downx + a.fw - 1, a.upx), a.in_w - 1);
    const int ww = ix_hi - ix_lo + 1, wh = iy_hi - iy_lo + 1;  // may be <= 0 (tile entirely in the padding)
    const int pitch = WIN_W + 1;
    // stage filter (already flipped as needed, gain folded in) and the window
    for (int i = threadIdx.x; i < a.fh * a.fw; i += 256) {
        const int ky = i / a.fw, kx = i % a.fw;
        const int fy = a.flip ? ky : a.fh - 1 - ky, fx = a.flip ? kx : a.fw - 1 - kx;
        s_f[ky * MAX_TAPS + kx] = a.f[fy * a.fs_h + fx * a.fs_w] * a.gain;
    }
    const T* xp = x + img * int64_t(a.in_h) * a.in_w;
    if (ww > 0 && wh > 0) {
        for (int i = threadIdx.x; i < wh * ww; i += 256) {
            const int r = i / ww, cidx = i % ww;
            s_x[r * pitch + cidx] = float(load_as<T>(xp, int64_t(iy_lo + r) * a.in_w + ix_lo + cidx));
        }
    }
    __syncthreads();
    const int lx = (threadIdx.x & 15) * OUT_PER_LANE, ly = threadIdx.x >> 4;
    if (ly >= th) return;
    const int oy = oy0 + ly;
    const int by = oy * a.downy - a.pady0;
    const int iy_a = max(ceil_div_pos(by, a.upy), 0), iy_b = min(floor_div_pos(by + a.fh - 1, a.upy), a.in_h - 1);
    A acc[OUT_PER_LANE];
#pragma unroll
    for (int q = 0; q < OUT_PER_LANE; q++) acc[q] = A(0);
    for (int iy = iy_a; iy <= iy_b; iy++) {
        const int ky = iy * a.upy - by;
        const float* frow = s_f + ky * MAX_TAPS;
        const float* xrow = s_x + (iy - iy_lo) * pitch - ix_lo;
#pragma unroll
        for (int q = 0; q < OUT_PER_LANE; q++) {
            const int ox = ox0 + lx + q;
            const int bx = ox * a.downx - a.padx0;
            const int ix_a = max(ceil_div_pos(bx, a.upx), 0), ix_b = min(floor_div_pos(bx + a.fw - 1, a.upx), a.in_w - 1);
            for (int ix = ix_a; ix <= ix_b; ix++) acc[q] += A(xrow[ix]) * A(frow[ix * a.upx - bx]);
        }
    }
    T* yp = y + img * int64_t(a.out_h) * a.out_w + int64_t(oy) * a.out_w + ox0 + lx;
#pragma unroll
    for (int q = 0; q < OUT_PER_LANE; q++)
        if (lx + q < tw) store_as<T>(yp, q, acc[q]);
}

// ---------------------------------------------------------------------------------------------
// Specialised kernel for what StyleGAN2 / G-NeRF actually issue: 4x4 filter, NCHW, same factor on both axes,
// (up,down) in {(1,1) blur, (2,1) upsample, (1,2) downsample} (and their gradients, which are the same shapes).
//   * a workgroup owns a TH-row x TW-column output tile of one image (Fir4Shape below); a lane makes OPL = 16 bytes of outputs
//   * the input window is staged into LDS as fp32 with ALIGNED 16-byte global loads (the window's rows start at
//     arbitrary element offsets -- 513-wide images -- so lanes walk the flat tensor in aligned vectors and mask);
//     out-of-image positions are staged as zeros, so the FIR loop has no bounds checks
//   * the polyphase structure is resolved at compile time: PX/PY = pad0 mod up fix which taps hit non-zero samples
//     for each of the lane's OPL outputs, so the inner loops are fully unrolled FMAs on registers
//   * one 16-byte store per lane.
template <class T> struct Vec16 { static constexpr int N = 16 / sizeof(T); };

// Tile shape.  A lane produces OPL horizontally adjacent outputs and reads a run of window columns that starts
// CS = OPL*DOWN/UP dwords after its left neighbour's.  With 16 lanes across, CS = 8 (fp16 blur: 2-way) or 16 (fp16
// down-sampling: 4-way) puts lanes of one row on the same LDS banks -- PMC showed 5x more bank-conflict cycles than LDS
// issue cycles for the fp16 blur, which ran at the same elements/s as fp32.  So the lanes across are limited to 64/CS
// and the tile grows downwards instead; with an odd row pitch the 64 lanes of a wave then touch 64 distinct banks.
template <class T, int UP, int DOWN> struct Fir4Shape {
    static constexpr int OPL = Vec16<T>::N;
    static constexpr int CS = (OPL * DOWN) / UP;
    static constexpr int LXN = CS >= 16 ? 4 : (CS >= 8 ? 8 : 16);    // lanes across the tile
    static constexpr int TW = LXN * OPL, TH = 256 / LXN;
};

template <class T, int UP, int DOWN, int PX, int PY>   // PY is implied by Y0 at run time; kept so launches are explicit about the phase
__global__ __launch_bounds__(256) void upfirdn_fir4_kernel(UpArgs a, int tiles_x, int tiles_y) {
    typedef Fir4Shape<T, UP, DOWN> Shape;
    constexpr int OPL = Shape::OPL, LXN = Shape::LXN, TW = Shape::TW, TH = Shape::TH, NT = 4 / UP;
    constexpr int WW = ((TW - 1) * DOWN + 3) / UP + 2, WH = ((TH - 1) * DOWN + 3) / UP + 2, PITCH = WW | 1;
    constexpr int SPAN = ((OPL - 1) * DOWN + (UP - 1)) / UP + NT;         // window columns one lane touches per row
    __shared__ float s_x[WH * PITCH];
    __shared__ float s_f[16];
    const T* x = static_cast<const T*>(a.x);
    T* y = static_cast<T*>(a.y);
    int t = blockIdx.x;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y; t /= tiles_y;
    const int64_t img = t;
    const int ox0 = tx * TW, oy0 = ty * TH;
    const int wx0 = floor_div_pos(ox0 * DOWN - a.padx0, UP), wy0 = floor_div_pos(oy0 * DOWN - a.pady0, UP);
    if (threadIdx.x < 16) {
        const int ky = threadIdx.x >> 2, kx = threadIdx.x & 3;
        const int fy = a.flip ? ky : 3 - ky, fx = a.flip ? kx : 3 - kx;
        s_f[threadIdx.x] = a.f[fy * a.fs_h + fx * a.fs_w] * a.gain;
    }
    // ---- stage the window
    const int64_t img_base = img * int64_t(a.in_h) * a.in_w, numel = int64_t(a.n) * a.c * a.in_h * a.in_w;
    constexpr int NV = (WW + OPL - 1) / OPL + 1;                          // aligned vectors that can overlap one window row
    for (int i = threadIdx.x; i < WH * NV; i += 256) {
        const int r = i / NV, v = i % NV;
        const int iy = wy0 + r;
        const int64_t row0 = img_base + int64_t(iy) * a.in_w + wx0;       // flat index of window column 0 (may be "virtual")
        const int64_t vec0 = ((row0 >= 0 ? row0 : row0 - (OPL - 1)) / OPL + v) * OPL;      // aligned vector start (floor)
        const bool row_ok = iy >= 0 && iy < a.in_h;
        T vals[OPL];
        const bool load_ok = row_ok && vec0 >= 0 && vec0 + OPL <= numel;
        if (load_ok) {
            *reinterpret_cast<Pack16<T>*>(vals) = *reinterpret_cast<const Pack16<T>*>(x + vec0);
        }
#pragma unroll
        for (int e = 0; e < OPL; e++) {
            const int pos = int(vec0 + e - row0);                         // window column of this element
            if (pos >= 0 && pos < WW) {
                const int ix = wx0 + pos;
                float val = 0.f;
                if (row_ok && ix >= 0 && ix < a.in_w) val = load_ok ? float(load_as<T>(vals, e)) : float(load_as<T>(x, vec0 + e));
                s_x[r * PITCH + pos] = val;
            }
        }
    }
    __syncthreads();
    // ---- FIR on registers
    const int lx = threadIdx.x % LXN, ly = threadIdx.x / LXN;
    const int oy = oy0 + ly, oxb = ox0 + OPL * lx;
    // rows: Y0 = oy*DOWN - pady0 is the upsampled row of tap 0; the first tap on a real sample is k0y = (-Y0) mod UP
    const int Y0 = oy * DOWN - a.pady0;
    const int k0y = (UP == 1) ? 0 : ((Y0 & 1) ? 1 : 0);
    const int row_base = ((Y0 + k0y) >> (UP == 2 ? 1 : 0)) - wy0;         // exact division
    float acc[OPL];
#pragma unroll
    for (int q = 0; q < OPL; q++) acc[q] = 0.f;
    const int X0 = oxb * DOWN - a.padx0;                                  // q = 0; oxb is a multiple of OPL (even)
    constexpr int k0x0 = (UP == 1) ? 0 : (PX & 1);                        // k0x for q = 0: (-X0) mod 2 = padx0 mod 2
    const int col_base = ((X0 + k0x0) >> (UP == 2 ? 1 : 0)) - wx0;
#pragma unroll
    for (int tyy = 0; tyy < NT; tyy++) {
        const float* xrow = s_x + (row_base + tyy) * PITCH + col_base;
        float xr[SPAN];
#pragma unroll
        for (int i = 0; i < SPAN; i++) xr[i] = xrow[i];
        const int ky = k0y + tyy * UP;
        float fr[4];
#pragma unroll
        for (int i = 0; i < 4; i++) fr[i] = s_f[ky * 4 + i];
#pragma unroll
        for (int q = 0; q < OPL; q++) {
            const int k0x = (UP == 1) ? 0 : ((k0x0 + q * DOWN) & 1);      // compile-time after unrolling
            const int rel = (q * DOWN + k0x - k0x0) / UP;                 // window column of tap 0 relative to col_base
#pragma unroll
            for (int txx = 0; txx < NT; txx++) acc[q] = fmaf(xr[rel + txx], fr[k0x + txx * UP], acc[q]);
        }
    }
    if (oy < a.out_h) {
        T* yp = y + img * int64_t(a.out_h) * a.out_w + int64_t(oy) * a.out_w + oxb;
        if (oxb + OPL <= a.out_w && ((reinterpret_cast<uintptr_t>(yp) & 15) == 0)) {
            T outv[OPL];
#pragma unroll
            for (int q = 0; q < OPL; q++) store_as<T>(outv, q, acc[q]);
            *reinterpret_cast<Pack16<T>*>(yp) = *reinterpret_cast<Pack16<T>*>(outv);
        } else {
#pragma unroll
            for (int q = 0; q < OPL; q++)
                if (oxb + q < a.out_w) store_as<T>(yp, q, acc[q]);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Streaming kernel for the 4x4 filter WITHOUT resampling (up = down = 1): the blur after every transposed
// convolution of the synthesis / superresolution blocks (conv2d_resample.py:114-131), the largest tensors of a forward
// pass ([N,128,513,513] -> 512).  No LDS and no barriers: a lane owns OPL = 16 bytes of adjacent output columns and walks
// down a strip of rows, holding the last four input rows (OPL + 3 values each) in registers; every step loads ONE new
// input row straight from global memory -- two unaligned vector loads per lane (the hardware takes 2-byte-aligned
// dwordx4 loads; rows of a 513-wide fp16 image start on odd elements), coalesced across the wave -- and makes one
// output row with 16 FMAs per output.
// Compared with the tile kernel above (stage through LDS, ~38 VALU + 230 SALU instructions and 5.5 LDS reads per
// output): ~20 VALU per output, input re-read factor (R+3)/R, so the kernel is HBM-bound instead of instruction-bound.
// Two refinements of the second session (profiles/r02_ops_*):
//  * Strips of one image ALTERNATE direction: even strips walk down, odd strips walk up.  The four waves of a workgroup own four
//    consecutive strips; walking all of them downwards, a strip's three halo rows are read at its START while the neighbour that
//    owns them reads them at its END, a whole strip later, when they have long left the L2: measured 11 % more HBM reads than the
//    algorithm needs.  With alternating directions both sides of every strip boundary inside a workgroup are read at the same
//    moment (both strips start there, or both end there), so the second read hits the L2.  An upward walk is the downward walk of
//    the vertically mirrored problem: rows addressed as H-1-row, filter rows reversed, pady0 -> 3 - (in_h - out_h + pady0).
//  * float16: the products are formed by v_dot2_f32_f16 on the packed halves as loaded (two taps per instruction, fp32
//    accumulation: exact products, the reference's float accumulation) instead of 16 conversions + 16 FMAs per output -- the
//    kernel was instruction-bound in fp16 (half the bytes per output, the same instruction count).  Taps are split hi + lo into
//    two halves each, so any fp32 filter is applied to ~2^-22; the [1,3,3,1] filters of the networks have lo = 0 and that half
//    of the work is skipped by a wave-uniform branch.  Odd-aligned tap pairs come from one v_alignbit per dword per input row.
template <class T, int RB>                                  // RB = rows per wait (see below)
__global__ __launch_bounds__(256) void upfirdn_blur4_kernel(UpArgs a, int lx_shift, int strips_x, int strips_y, int rows_per_strip) {
    constexpr bool kHalf = sizeof(T) == 2;
    constexpr int OPL = 16 / sizeof(T), NIN = OPL + 3;
    constexpr int NLOAD = kHalf ? 12 : 8;                   // elements fetched per row: dwordx4 + dwordx2 (fp16) / dwordx4 x2 (fp32)
    constexpr int ND = NLOAD * sizeof(T) / 4;               // ... as dwords
    constexpr int NROW = kHalf ? 11 : NIN;                  // registers per input row: 6 aligned + 5 odd-aligned pairs (fp16) / 7 floats (fp32)
    // explicitly under-aligned vector types: ONE global_load_dwordx4 / dwordx2 each (a memcpy from a 2-byte-aligned
    // address is split by the compiler into dwordx3 + dwordx2 + ushort pieces)
    typedef unsigned u4u __attribute__((ext_vector_type(4), aligned(2)));
    typedef unsigned u2u __attribute__((ext_vector_type(2), aligned(2)));
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const T* x = static_cast<const T*>(a.x);
    T* y = static_cast<T*>(a.y);
    const int lane = threadIdx.x & 63;
    const int64_t gw = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
    const int lx = lane & ((1 << lx_shift) - 1), sub = lane >> lx_shift;
    int64_t sid = (gw << (6 - lx_shift)) + sub;              // strip id: (image, strip row, strip column)
    const int sx = int(sid % strips_x); sid /= strips_x;
    const int sy = int(sid % strips_y); sid /= strips_y;
    const int64_t img = sid;
    const bool live = img < int64_t(a.n) * a.c;
    // everything below works in the strip's WALK coordinates: rows count from the top for a downward walk, from the bottom for an
    // upward one (odd strips)
    const bool up = (sy & 1) != 0;
    const int pady0 = up ? 3 - (a.in_h - a.out_h + a.pady0) : a.pady0;
    const int rows_here = min(rows_per_strip, a.out_h - sy * rows_per_strip);
    const int ox0 = ((sx << lx_shift) + lx) * OPL, oy0 = up ? a.out_h - (sy * rows_per_strip + rows_here) : sy * rows_per_strip;
    const int ix0 = ox0 - a.padx0;
    unsigned col_ok = 0;
#pragma unroll
    for (int e = 0; e < NIN; e++) col_ok |= (live && ix0 + e >= 0 && ix0 + e < a.in_w) ? (1u << e) : 0u;
    // filter taps, flipped like a true convolution unless a.flip, gain folded in (wave-uniform: scalar registers)
    float f[4][4];
#pragma unroll
    for (int ky = 0; ky < 4; ky++) {
#pragma unroll
        for (int kx = 0; kx < 4; kx++) {
            const int kyp = up ? 3 - ky : ky;
            f[ky][kx] = a.f[(a.flip ? kyp : 3 - kyp) * a.fs_h + (a.flip ? kx : 3 - kx) * a.fs_w] * a.gain;
        }
    }
    // fp16: taps as packed pairs, split hi + lo
    h2 fh[4][2], fl[4][2];
    bool has_lo = false;
    unsigned cmask[6] = {0, 0, 0, 0, 0, 0};
    if constexpr (kHalf) {
#pragma unroll
        for (int ky = 0; ky < 4; ky++) {
#pragma unroll
            for (int p2 = 0; p2 < 2; p2++) {
                const _Float16 h0 = (_Float16)f[ky][2 * p2], h1 = (_Float16)f[ky][2 * p2 + 1];
                const _Float16 l0 = (_Float16)(f[ky][2 * p2] - (float)h0), l1 = (_Float16)(f[ky][2 * p2 + 1] - (float)h1);
                fh[ky][p2] = (h2){h0, h1};
                fl[ky][p2] = (h2){l0, l1};
                has_lo = has_lo || (float)l0 != 0.f || (float)l1 != 0.f;
            }
        }
#pragma unroll
        for (int i = 0; i < 6; i++)
            cmask[i] = (((col_ok >> (2 * i)) & 1u) ? 0xffffu : 0u) | ((2 * i + 1 < NIN && ((col_ok >> (2 * i + 1)) & 1u)) ? 0xffff0000u : 0u);
    }
    const int64_t numel = int64_t(a.n) * a.c * a.in_h * a.in_w;

    // Loads are issued unconditionally from an address clamped into the tensor -- a branch around them would make the
    // compiler wait for the data at the end of the branch (observed: s_waitcnt right behind the load, no overlap with the
    // previous row's arithmetic).  Rows outside the image and edge columns are zeroed when the row is unpacked; the few
    // lanes whose true window leaves the tensor (its first / last elements) re-read it element-wise there.
    const int64_t img_c = live ? img : 0;
    const int64_t img_base_c = img_c * int64_t(a.in_h) * a.in_w;
    auto row_index = [&](int iy) -> int64_t {               // iy in walk coordinates
        const int phys = up ? a.in_h - 1 - iy : iy;
        return img_base_c + int64_t(min(max(phys, 0), a.in_h - 1)) * a.in_w + ix0;
    };
    struct Raw { unsigned d[ND]; };
    auto fetch = [&](int iy, Raw& raw) {
        const int64_t idx = min(max(row_index(iy), int64_t(0)), numel - NLOAD);
        const u4u lo = *reinterpret_cast<const u4u*>(x + idx);
        raw.d[0] = lo[0]; raw.d[1] = lo[1]; raw.d[2] = lo[2]; raw.d[3] = lo[3];
        if constexpr (kHalf) {
            const u2u hi = *reinterpret_cast<const u2u*>(x + idx + 8);
            raw.d[4] = hi[0]; raw.d[5] = hi[1];
        } else {
            const u4u hi = *reinterpret_cast<const u4u*>(x + idx + 4);
            raw.d[4] = hi[0]; raw.d[5] = hi[1]; raw.d[6] = hi[2]; raw.d[7] = hi[3];
        }
    };
    // One input row in registers.  fp32: NIN floats.  fp16: dwords 0..5 = element pairs (0,1) (2,3) ... (10,-) masked, dwords 6..10 =
    // the odd-aligned pairs (1,2) (3,4) ... (9,10).
    struct Row { unsigned r[NROW]; };
    Row rows[3 + RB];
    auto unpack = [&](int iy, const Raw& raw, Row& dst) {
        const bool row_ok = iy >= 0 && iy < a.in_h;
        const unsigned ok = row_ok ? col_ok : 0u;
        const int64_t idx = row_index(iy);
        const bool ragged = ok != 0 && (idx < 0 || idx + NLOAD > numel);     // first / last elements of the tensor: element-wise re-read
        if constexpr (kHalf) {
            unsigned d[6];
#pragma unroll
            for (int i = 0; i < 6; i++) d[i] = row_ok ? (raw.d[i] & cmask[i]) : 0u;
            if (ragged) {
#pragma unroll
                for (int i = 0; i < 6; i++) {
                    unsigned lo = 0, hi = 0;
                    if (((ok >> (2 * i)) & 1u) && idx + 2 * i >= 0 && idx + 2 * i < numel) lo = __half_as_ushort(x[idx + 2 * i]);
                    if (2 * i + 1 < NIN && ((ok >> (2 * i + 1)) & 1u) && idx + 2 * i + 1 >= 0 && idx + 2 * i + 1 < numel) hi = __half_as_ushort(x[idx + 2 * i + 1]);
                    d[i] = lo | (hi << 16);
                }
            }
#pragma unroll
            for (int i = 0; i < 6; i++) dst.r[i] = d[i];
#pragma unroll
            for (int i = 0; i < 5; i++) dst.r[6 + i] = __builtin_amdgcn_alignbit(d[i + 1], d[i], 16);      // (element 2i+1, element 2i+2)
        } else {
#pragma unroll
            for (int e = 0; e < NIN; e++) dst.r[e] = ((ok >> e) & 1u) ? raw.d[e] : 0u;
            if (ragged) {
#pragma unroll
                for (int e = 0; e < NIN; e++)
                    if ((ok >> e) & 1u) dst.r[e] = (idx + e >= 0 && idx + e < numel) ? __float_as_uint(float(load_as<T>(x, idx + e))) : 0u;
            }
        }
    };
    auto emit = [&](int oy, const Row& r0, const Row& r1, const Row& r2, const Row& r3) {        // oy in walk coordinates
        if (!live || oy >= oy0 + rows_here || ox0 >= a.out_w) return;
        float acc[OPL];
        const Row* rr[4] = {&r0, &r1, &r2, &r3};
        if constexpr (kHalf) {
#pragma unroll
            for (int q = 0; q < OPL; q++) {
                float sacc = 0.f;
#pragma unroll
                for (int ky = 0; ky < 4; ky++) {
                    // outputs at even q read aligned pairs (q,q+1),(q+2,q+3); odd q the odd-aligned ones
                    const unsigned pa = (q & 1) ? rr[ky]->r[6 + (q >> 1)] : rr[ky]->r[q >> 1];
                    const unsigned pb = (q & 1) ? rr[ky]->r[6 + (q >> 1) + 1] : rr[ky]->r[(q >> 1) + 1];
                    sacc = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, pa), fh[ky][0], sacc, false);
                    sacc = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, pb), fh[ky][1], sacc, false);
                }
                acc[q] = sacc;
            }
            if (has_lo) {                                    // wave-uniform: taps that are not exact in fp16
#pragma unroll
                for (int q = 0; q < OPL; q++) {
#pragma unroll
                    for (int ky = 0; ky < 4; ky++) {
                        const unsigned pa = (q & 1) ? rr[ky]->r[6 + (q >> 1)] : rr[ky]->r[q >> 1];
                        const unsigned pb = (q & 1) ? rr[ky]->r[6 + (q >> 1) + 1] : rr[ky]->r[(q >> 1) + 1];
                        acc[q] = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, pa), fl[ky][0], acc[q], false);
                        acc[q] = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, pb), fl[ky][1], acc[q], false);
                    }
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < OPL; q++) {
                float sacc = 0.f;
#pragma unroll
                for (int ky = 0; ky < 4; ky++) {
#pragma unroll
                    for (int kx = 0; kx < 4; kx++) sacc = fmaf(__uint_as_float(rr[ky]->r[q + kx]), f[ky][kx], sacc);
                }
                acc[q] = sacc;
            }
        }
        const int oy_phys = up ? a.out_h - 1 - oy : oy;
        T* yp = y + img * int64_t(a.out_h) * a.out_w + int64_t(oy_phys) * a.out_w + ox0;
        if (ox0 + OPL <= a.out_w && ((reinterpret_cast<uintptr_t>(yp) & 15) == 0)) {
            uint4 w;                                         // ONE global_store_dwordx4
            if constexpr (kHalf) {
                w.x = __builtin_bit_cast(unsigned, (h2){(_Float16)acc[0], (_Float16)acc[1]});
                w.y = __builtin_bit_cast(unsigned, (h2){(_Float16)acc[2], (_Float16)acc[3]});
                w.z = __builtin_bit_cast(unsigned, (h2){(_Float16)acc[4 % OPL], (_Float16)acc[5 % OPL]});
                w.w = __builtin_bit_cast(unsigned, (h2){(_Float16)acc[6 % OPL], (_Float16)acc[7 % OPL]});
            } else {
                w = make_uint4(__float_as_uint(acc[0]), __float_as_uint(acc[1]), __float_as_uint(acc[2]), __float_as_uint(acc[3]));
            }
            *reinterpret_cast<uint4*>(yp) = w;
        } else {
#pragma unroll
            for (int q = 0; q < OPL; q++)
                if (ox0 + q < a.out_w) store_as<T>(yp, q, acc[q]);
        }
    };
    // Rows are processed in BLOCKS OF RB = FOUR.  On this hardware loads and stores share one counter (vmcnt), and with both
    // kinds pending a wait for the loads also waits for every earlier store to be acknowledged; with one row per wait the
    // kernel ran at 3 TB/s no matter how the arithmetic or the prefetch was arranged (removing the stores alone doubled
    // its speed).  So: one wait per block -- it drains the previous block's four stores and this block's four row loads,
    // which were both in flight during the previous block's arithmetic -- then the next block's loads are issued, then
    // four output rows are made and stored.  rows[0..2] are the last three input rows of the previous block.
    const int iy_first = oy0 - pady0;
    Raw raw[RB < 3 ? 3 : RB];
#pragma unroll
    for (int u = 0; u < 3; u++) fetch(iy_first + u, raw[u]);
#pragma unroll
    for (int u = 0; u < 3; u++) unpack(iy_first + u, raw[u], rows[u]);
#pragma unroll
    for (int u = 0; u < RB; u++) fetch(iy_first + 3 + u, raw[u]);
    for (int r0 = 0; r0 < rows_here; r0 += RB) {
#pragma unroll
        for (int u = 0; u < RB; u++) unpack(iy_first + 3 + r0 + u, raw[u], rows[3 + u]);
#pragma unroll
        for (int u = 0; u < RB; u++) fetch(iy_first + 3 + RB + r0 + u, raw[u]);
#pragma unroll
        for (int u = 0; u < RB; u++) emit(oy0 + r0 + u, rows[u], rows[u + 1], rows[u + 2], rows[u + 3]);
#pragma unroll
        for (int u = 0; u < 3; u++) rows[u] = rows[RB + u];
    }
}

template <class T>
int launch_blur4(const UpArgs& a, hipStream_t stream) {
    constexpr int OPL = 16 / sizeof(T);
    if (int64_t(a.n) * a.c * a.in_h * a.in_w < 16) return 1;          // the clamped row loads need 12 elements to exist
    const int lanes_needed = (a.out_w + OPL - 1) / OPL;
    int lx_shift = 0;
    while (lx_shift < 6 && (1 << lx_shift) < lanes_needed) lx_shift++;
    const int strips_x = (lanes_needed + (1 << lx_shift) - 1) >> lx_shift;
    // rows per wait: 4 for fp32 (1, 2, 3, 6, 8 measured slower); 3 for fp16, whose packed row window then fits 128 VGPRs = 4 waves
    // per SIMD (with 4 the kernel needs 130 and drops to 3: 4.04 vs ... TB/s).  GNERF_BLUR_RB overrides for A/B runs.
    int rb = sizeof(T) == 2 ? 3 : 4;
    if (const char* e = getenv("GNERF_BLUR_RB")) rb = atoi(e);
    const int rows_per_strip = (a.out_h > 256 ? 32 : 16) / rb * rb + (rb == 3 ? 3 : 0);      // a multiple of rb: 32/16 (rb 2, 4), 33/18 (rb 3)
    if (a.in_h - a.out_h + a.pady0 > 3 || a.in_h - a.out_h + a.pady0 < 0) return 1;   // upward walks need the mirrored padding inside the filter
    const int strips_y = (a.out_h + rows_per_strip - 1) / rows_per_strip;
    const int64_t strips = int64_t(strips_x) * strips_y * a.n * a.c;
    const int64_t waves = (strips + (64 >> lx_shift) - 1) / (64 >> lx_shift);
    const int64_t blocks = (waves + 3) / 4;
    if (blocks > INT32_MAX) return 1;
    if (rb == 2)      hipLaunchKernelGGL((upfirdn_blur4_kernel<T, 2>), dim3((unsigned)blocks), dim3(256), 0, stream, a, lx_shift, strips_x, strips_y, rows_per_strip);
    else if (rb == 3) hipLaunchKernelGGL((upfirdn_blur4_kernel<T, 3>), dim3((unsigned)blocks), dim3(256), 0, stream, a, lx_shift, strips_x, strips_y, rows_per_strip);
    else              hipLaunchKernelGGL((upfirdn_blur4_kernel<T, 4>), dim3((unsigned)blocks), dim3(256), 0, stream, a, lx_shift, strips_x, strips_y, rows_per_strip);
    return check_launch("upfirdn2d(blur4)") == GNERF_OK ? 0 : -1;
}

// ---------------------------------------------------------------------------------------------
// The same 4x4 blur for CHANNELS_LAST tensors (memory [N,H,W,C]): what the fp16 blocks hold when they run in the layout
// MIOpen's fp16 kernels compute in (no NCHW<->NHWC transposes around every convolution: tools/bench_sr_conv_layout.py), or when
// the reference's `fp16_channels_last` is set (networks_stylegan2.py:385,417).  A lane owns one 16-byte channel vector of one
// output column and walks down a strip of rows; lanes run along (column, channel vector) = along memory, so every access is
// a coalesced 16-byte vector.  Formulated as a SCATTER over input rows: input row r contributes to the four output rows
// r-3+pad .. r+pad, so the lane keeps four output accumulators (fp32) instead of a 4x4 window of inputs; each step loads the
// four column vectors of one input row (issued one row ahead), converts them once, adds them into the four pending rows and
// stores the row that is complete.  Input re-read (R+3)/R over the strip boundary, served by the L2 for neighbouring strips.
// EPI: the modulated convolution's epilogue runs on the blurred value before it is stored (the x2-upsampling layers: transposed
// convolution -> this blur -> demodulation + bias + lrelu + clamp [+ the next layer's input scale]; one pass over the activations
// instead of two).  The blurred value is rounded to T first, as the separate blur would have stored it, so the result is bit-identical
// to blur followed by gnerf_modconv_epilogue_nhwc.
struct BlurEpi {
    const float* scale;        // [n, c] demodulation coefficients or NULL
    const void* bias;          // [c] in T or NULL
    const float* next_scale;   // [n, c] or NULL
    float alpha, gain, clamp;
};

template <class T, int EPI>                                 // EPI: 0 none, 1 linear, 3 lrelu
__global__ __launch_bounds__(256) void upfirdn_blur4_nhwc_kernel(UpArgs a, int cv, int strips_y, int rows_per_strip, BlurEpi ep) {
    constexpr int VEC = 16 / sizeof(T);
    // grid = (column blocks, strips, images): strip and image are the workgroup's, so the row walk below (steps, row validity, row
    // addresses) is wave-uniform -- with one flat index they were per-lane values, every `if (s < steps)` became an exec-masked
    // region, and the vectors loaded for the next row passed through register copies (and a wait for everything in flight) at each
    // Workgroups are handed to the eight XCDs round-robin in launch order, and each XCD has its own L2: column-neighbours -- which share
    // three of every 4..16 input columns -- and strip-neighbours would sit on different XCDs and each fetch the shared lines from HBM
    // (FETCH x 2 + WRITE 1.13-1.25 x the algorithmic bytes).  Re-number so that every XCD owns a contiguous range of workgroups.
    unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    {
        const unsigned total = gridDim.x * gridDim.y * gridDim.z;
        if (total % kNumXCD == 0) {
            const unsigned lin = bx + gridDim.x * (by + gridDim.y * bz);
            const unsigned re = (lin % kNumXCD) * (total / kNumXCD) + lin / kNumXCD;
            bx = re % gridDim.x; by = (re / gridDim.x) % gridDim.y; bz = re / (gridDim.x * gridDim.y);
        }
    }
    const int64_t per_row = int64_t(a.out_w) * cv;
    const int64_t t = int64_t(bx) * 256 + threadIdx.x;
    if (t >= per_row) return;
    const int col = int(t);
    const int sy = by;
    const int img = bz;
    const int ox = col / cv, c0 = (col % cv) * VEC;
    const int oy0 = sy * rows_per_strip, rows = min(rows_per_strip, a.out_h - oy0);
    const T* x = static_cast<const T*>(a.x) + img * a.xs_n + c0;
    T* y = static_cast<T*>(a.y) + img * a.ys_n + int64_t(ox) * a.ys_w + c0;
    // taps in walk order: tap[ky][kx] multiplies input (oy - pady0 + ky, ox - padx0 + kx)
    float tap[4][4];
#pragma unroll
    for (int ky = 0; ky < 4; ky++)
#pragma unroll
        for (int kx = 0; kx < 4; kx++)
            tap[ky][kx] = a.f[(a.flip ? ky : 3 - ky) * a.fs_h + (a.flip ? kx : 3 - kx) * a.fs_w];
    // (Measured and dropped: applying the rank-one binomial filter as a row pass + a column pass -- 8 FMAs per value and input row instead
    // of 16, detected from the taps on the device -- made the plain blur SLOWER, 4.4 -> 3.75 TB/s, and left the fused form at 2.6: 150
    // VGPRs instead of 137 and a dependent chain where the 2-D form has four independent ones.  The kernel is not bound by its FMA count.)
    // float16 (round 5): the taps as packed halves for v_dot2_f32_f16 -- two COLUMNS of one channel per instruction, repacked from the
    // loaded (channel, channel + 1) dwords by v_perm_b32 -- instead of 32 conversions + 64 packed fp32 FMAs per input row: the kernel is
    // bound by its vector instructions, and this is the blur of csrc's NCHW kernel, which has formed its products this
    // way since round 2.  Taps are split hi + lo, so any fp32 filter is applied to ~2^-22; the networks' [1,3,3,1] filters have lo = 0
    // and that half is skipped by a wave-uniform branch.  The plain and the fused form are the same instantiation family: still bit-identical.
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    h2 th[4][2], tl[4][2];
    bool has_lo = false;
    if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int ky = 0; ky < 4; ky++)
#pragma unroll
            for (int kp = 0; kp < 2; kp++) {
                const float t0 = tap[ky][2 * kp], t1 = tap[ky][2 * kp + 1];
                const _Float16 h0 = (_Float16)t0, h1 = (_Float16)t1;
                const _Float16 l0 = (_Float16)(t0 - (float)h0), l1 = (_Float16)(t1 - (float)h1);
                th[ky][kp] = (h2){h0, h1};
                tl[ky][kp] = (h2){l0, l1};
                has_lo = has_lo || (float)l0 != 0.f || (float)l1 != 0.f;
            }
    }
    const int ix0 = ox - a.padx0;
    bool col_ok[4];
    int64_t col_off[4];
#pragma unroll
    for (int kx = 0; kx < 4; kx++) {
        col_ok[kx] = ix0 + kx >= 0 && ix0 + kx < a.in_w;
        col_off[kx] = int64_t(min(max(ix0 + kx, 0), a.in_w - 1)) * a.xs_w;          // clamped: loads are unconditional, values masked
    }
    // this lane's channel vector of the epilogue operands, fetched as whole vectors under uniform branches: element-by-element
    // conditional loads compiled to 24 dependent round trips in front of every strip (a global_load_dword + s_waitcnt vmcnt(0) each)
    float e_sc[VEC], e_bv[VEC], e_nx[VEC];
#pragma unroll
    for (int k = 0; k < VEC; k++) { e_sc[k] = 1.f; e_nx[k] = 1.f; e_bv[k] = 0.f; }
    if constexpr (EPI != 0) {
        float4 sv[VEC / 4], nv[VEC / 4];
        Pk<T, VEC> bv;
        const bool has_s = ep.scale != nullptr, has_n = ep.next_scale != nullptr, has_b = ep.bias != nullptr;
        if (has_s) {
#pragma unroll
            for (int q = 0; q < VEC / 4; q++) sv[q] = *reinterpret_cast<const float4*>(ep.scale + int64_t(img) * a.c + c0 + 4 * q);
        }
        if (has_n) {
#pragma unroll
            for (int q = 0; q < VEC / 4; q++) nv[q] = *reinterpret_cast<const float4*>(ep.next_scale + int64_t(img) * a.c + c0 + 4 * q);
        }
        if (has_b) bv = *reinterpret_cast<const Pk<T, VEC>*>(static_cast<const T*>(ep.bias) + c0);
        if (has_s) {
#pragma unroll
            for (int q = 0; q < VEC / 4; q++) { e_sc[4 * q] = sv[q].x; e_sc[4 * q + 1] = sv[q].y; e_sc[4 * q + 2] = sv[q].z; e_sc[4 * q + 3] = sv[q].w; }
        }
        if (has_n) {
#pragma unroll
            for (int q = 0; q < VEC / 4; q++) {
                e_nx[4 * q] = round_to<T>(nv[q].x); e_nx[4 * q + 1] = round_to<T>(nv[q].y); e_nx[4 * q + 2] = round_to<T>(nv[q].z); e_nx[4 * q + 3] = round_to<T>(nv[q].w);
            }
        }
        if (has_b) {
#pragma unroll
            for (int k = 0; k < VEC; k++) e_bv[k] = float(load_as<T>(bv.v, k));
        }
    }
    const int iy_first = oy0 - a.pady0;
    // fp32 arithmetic on PAIRS of channels (v_pk_fma_f32: the kernel is instruction-bound otherwise -- 128 scalar FMAs + 39 selects
    // per output vector ran at 3.6 TB/s in fp16); elements outside the image are zeroed on the raw dwords (data, not taps:
    // a clamped load may have fetched an Inf that the reference's zero padding never sees).
    typedef float f2 __attribute__((ext_vector_type(2)));
    constexpr int DW = 16 / 4;                                      // dwords per vector
    struct alignas(16) Raw { unsigned d[DW]; };
    f2 acc[4][VEC / 2];                                             // (fp32 tensors)
    float hacc[4][VEC];                                             // (fp16 tensors: one fp32 accumulator per channel)
    Raw raw[4];
    auto fetch_raw = [&](int iy, Raw (&dst)[4]) {
        const int64_t row = int64_t(min(max(iy, 0), a.in_h - 1)) * a.xs_h;
#pragma unroll
        for (int kx = 0; kx < 4; kx++) dst[kx] = *reinterpret_cast<const Raw*>(x + row + col_off[kx]);
    };
    fetch_raw(iy_first, raw);
    const int steps = rows + 3;
    // Order of a step, set by the ONE completion counter loads and stores share (vmcnt; the two kinds complete out of order with respect
    // to each other, so a wait for a load with a store in flight is a wait for everything): [wait] convert the row that was loaded a
    // step ago -> store the output row finished a step ago -> issue the next row's loads -> arithmetic -> epilogue into `pend`.
    // Loads and the store are then both a whole step old when the next step waits.  (Storing a row as soon as it was finished put the
    // store a few instructions in front of that wait: every step paid a store's round trip -- 2.6 TB/s fused, 4.4 plain.)
    Pk<T, VEC> pend;
    int pend_row = -1;
    for (int s4 = 0; s4 < steps; s4 += 4) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int s = s4 + u;                                   // input row iy_first + s
            if (s < steps) {                                        // (uniform over the strip's lanes)
                const bool row_ok = iy_first + s >= 0 && iy_first + s < a.in_h;
                if constexpr (sizeof(T) == 2) {
                    unsigned p01[VEC], p23[VEC];                    // (column 0, column 1) and (column 2, column 3) of channel c
                    {
                        unsigned m[4][DW];
#pragma unroll
                        for (int kx = 0; kx < 4; kx++) {
                            unsigned keep = (row_ok && col_ok[kx]) ? ~0u : 0u;
                            asm volatile("" : "+v"(keep));            // keep it a mask: one AND per dword
#pragma unroll
                            for (int q = 0; q < DW; q++) m[kx][q] = raw[kx].d[q] & keep;
                        }
#pragma unroll
                        for (int q = 0; q < DW; q++) {
                            p01[2 * q]     = __builtin_amdgcn_perm(m[1][q], m[0][q], 0x05040100u);
                            p01[2 * q + 1] = __builtin_amdgcn_perm(m[1][q], m[0][q], 0x07060302u);
                            p23[2 * q]     = __builtin_amdgcn_perm(m[3][q], m[2][q], 0x05040100u);
                            p23[2 * q + 1] = __builtin_amdgcn_perm(m[3][q], m[2][q], 0x07060302u);
                        }
                    }
                    // (pin the repacked values in front of the store: left alone, the compiler sinks the masking below it, and the wait
                    // for the loaded row then follows the store it must not wait for)
#pragma unroll
                    for (int c = 0; c < VEC; c += 4)
                        asm volatile("" : "+v"(p01[c]), "+v"(p01[c + 1]), "+v"(p01[c + 2]), "+v"(p01[c + 3]), "+v"(p23[c]), "+v"(p23[c + 1]), "+v"(p23[c + 2]), "+v"(p23[c + 3]) :: "memory");
                    if (pend_row >= 0) *reinterpret_cast<Pk<T, VEC>*>(y + int64_t(pend_row) * a.ys_h) = pend;
                    fetch_raw(iy_first + s + 1, raw);               // (past the last step: a clamped, unused row -- no branch around the loads)
                    __builtin_amdgcn_sched_barrier(0);
                    // this row is tap row ky of output row s - ky; accumulator (s - ky) & 3 = (u - ky) & 3
#pragma unroll
                    for (int ky = 0; ky < 4; ky++) {
                        float (&ac)[VEC] = hacc[(u - ky) & 3];
#pragma unroll
                        for (int c = 0; c < VEC; c++) {
                            float v = ky == 0 ? 0.f : ac[c];
                            v = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, p01[c]), th[ky][0], v, false);
                            v = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, p23[c]), th[ky][1], v, false);
                            ac[c] = v;
                        }
                    }
                    if (has_lo) {                                   // wave-uniform: taps that are not exact in fp16
#pragma unroll
                        for (int ky = 0; ky < 4; ky++) {
                            float (&ac)[VEC] = hacc[(u - ky) & 3];
#pragma unroll
                            for (int c = 0; c < VEC; c++) {
                                ac[c] = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, p01[c]), tl[ky][0], ac[c], false);
                                ac[c] = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, p23[c]), tl[ky][1], ac[c], false);
                            }
                        }
                    }
                } else {
                f2 in[4][VEC / 2];
#pragma unroll
                for (int kx = 0; kx < 4; kx++) {
                    unsigned keep = (row_ok && col_ok[kx]) ? ~0u : 0u;
                    asm volatile("" : "+v"(keep));                     // keep it a mask: one AND per dword instead of one select per converted float
#pragma unroll
                    for (int q = 0; q < DW; q++) {
                        const unsigned d = raw[kx].d[q] & keep;
                        if constexpr (sizeof(T) == 2) {
                            typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                            const h2 hv = __builtin_bit_cast(h2, d);
                            in[kx][q] = (f2){(float)hv[0], (float)hv[1]};
                        } else {
                            if ((q & 1) == 0) in[kx][q / 2][0] = __uint_as_float(d); else in[kx][q / 2][1] = __uint_as_float(d);
                        }
                    }
                }
                // (pin the converted values in front of the store: left alone, the compiler sinks the masking below it, and the wait for
                // the loaded row then follows the store it must not wait for)
#pragma unroll
                for (int kx = 0; kx < 4; kx++) {
                    if constexpr (VEC == 8) asm volatile("" : "+v"(in[kx][0]), "+v"(in[kx][1]), "+v"(in[kx][2]), "+v"(in[kx][3]) :: "memory");
                    else                    asm volatile("" : "+v"(in[kx][0]), "+v"(in[kx][1]) :: "memory");
                }
                if (pend_row >= 0) *reinterpret_cast<Pk<T, VEC>*>(y + int64_t(pend_row) * a.ys_h) = pend;
                fetch_raw(iy_first + s + 1, raw);                   // (past the last step: a clamped, unused row -- no branch around the loads)
                __builtin_amdgcn_sched_barrier(0);
                // this row is tap row ky of output row s - ky; accumulator (s - ky) & 3 = (u - ky) & 3
#pragma unroll
                for (int ky = 0; ky < 4; ky++) {
                    f2 (&ac)[VEC / 2] = acc[(u - ky) & 3];
#pragma unroll
                    for (int k = 0; k < VEC / 2; k++) {
                        f2 v = ky == 0 ? (f2){0.f, 0.f} : ac[k];
#pragma unroll
                        for (int kx = 0; kx < 4; kx++) v = __builtin_elementwise_fma(in[kx][k], (f2){tap[ky][kx], tap[ky][kx]}, v);
                        ac[k] = v;
                    }
                }
                }
                const int done = s - 3;                             // output row oy0 + done is complete
                pend_row = -1;
                if (done >= 0) {
                    Pk<T, VEC> out;
#pragma unroll
                    for (int k = 0; k < VEC / 2; k++) {
                        f2 r;
                        if constexpr (sizeof(T) == 2) r = (f2){hacc[(u - 3) & 3][2 * k], hacc[(u - 3) & 3][2 * k + 1]} * (f2){a.gain, a.gain};
                        else                          r = acc[(u - 3) & 3][k] * (f2){a.gain, a.gain};
                        store_as<T>(out.v, 2 * k, r[0]);
                        store_as<T>(out.v, 2 * k + 1, r[1]);
                    }
                    if constexpr (EPI != 0) {
                        if (ep.scale && ep.next_scale) out = modconv_epilogue_vec<T, VEC, EPI, true, false, true>(out, e_sc, 0.f, false, e_bv, e_nx, ep.alpha, ep.gain, ep.clamp);
                        else if (ep.scale)             out = modconv_epilogue_vec<T, VEC, EPI, true, false, false>(out, e_sc, 0.f, false, e_bv, e_nx, ep.alpha, ep.gain, ep.clamp);
                        else if (ep.next_scale)        out = modconv_epilogue_vec<T, VEC, EPI, false, false, true>(out, e_sc, 0.f, false, e_bv, e_nx, ep.alpha, ep.gain, ep.clamp);
                        else                           out = modconv_epilogue_vec<T, VEC, EPI, false, false, false>(out, e_sc, 0.f, false, e_bv, e_nx, ep.alpha, ep.gain, ep.clamp);
                    }
                    pend = out;
                    pend_row = oy0 + done;
                }
            }
        }
    }
    if (pend_row >= 0) *reinterpret_cast<Pk<T, VEC>*>(y + int64_t(pend_row) * a.ys_h) = pend;
}

template <class T>
int launch_blur4_nhwc(const UpArgs& a, hipStream_t stream, const BlurEpi* ep = nullptr, int act = 0) {
    constexpr int VEC = 16 / sizeof(T);
    const int cv = a.c / VEC;
    // Strip length: long strips re-read little ((R + 3) / R input rows per output row), but a lane is one (column, channel vector) of one
    // strip, and ONE 512x512x64 image in 64-row strips is 512 waves -- two per CU (the orbit's frames: 50 us per call, 1.0 TB/s).  Take the
    // longest strip that still gives every CU 16 waves; the re-read of short strips is served by the L2.  (The result does not depend on
    // the strip length: an output row accumulates its four input rows in the same order wherever the strip starts.)
    int rows_per_strip = a.out_h > 256 ? 64 : (a.out_h > 64 ? 32 : 16);
    while (rows_per_strip > 8 && int64_t(a.out_w) * cv * ((a.out_h + rows_per_strip - 1) / rows_per_strip) * a.n < int64_t(kNumCU) * 16 * 64) rows_per_strip >>= 1;
    const int strips_y = (a.out_h + rows_per_strip - 1) / rows_per_strip;
    const int64_t blocks = (int64_t(a.out_w) * cv + 255) / 256;
    if (blocks > INT32_MAX || strips_y > 65535 || a.n > 65535) return 1;
    const dim3 g((unsigned)blocks, (unsigned)strips_y, (unsigned)a.n), b(256);
    const BlurEpi none = {nullptr, nullptr, nullptr, 0.f, 1.f, -1.f};
    if (!ep)           hipLaunchKernelGGL((upfirdn_blur4_nhwc_kernel<T, 0>), g, b, 0, stream, a, cv, strips_y, rows_per_strip, none);
    else if (act == 3 && ep->alpha >= 0.f && ep->alpha <= 1.f) hipLaunchKernelGGL((upfirdn_blur4_nhwc_kernel<T, kActLrelu01>), g, b, 0, stream, a, cv, strips_y, rows_per_strip, *ep);
    else if (act == 3) hipLaunchKernelGGL((upfirdn_blur4_nhwc_kernel<T, 3>), g, b, 0, stream, a, cv, strips_y, rows_per_strip, *ep);
    else               hipLaunchKernelGGL((upfirdn_blur4_nhwc_kernel<T, 1>), g, b, 0, stream, a, cv, strips_y, rows_per_strip, *ep);
    return check_launch("upfirdn2d(blur4, channels_last)") == GNERF_OK ? 0 : -1;
}

template <class T, int UP, int DOWN>
int launch_fir4(const UpArgs& a, hipStream_t stream) {
    constexpr int TW = Fir4Shape<T, UP, DOWN>::TW, TH = Fir4Shape<T, UP, DOWN>::TH;
    const int tiles_x = (a.out_w + TW - 1) / TW, tiles_y = (a.out_h + TH - 1) / TH;
    const int64_t blocks = int64_t(tiles_x) * tiles_y * a.n * a.c;
    if (blocks > INT32_MAX) return 1;
    const dim3 g((unsigned)blocks), b(256);
    const int px = ((a.padx0 % UP) + UP) % UP, py = ((a.pady0 % UP) + UP) % UP;
    if (UP == 1) hipLaunchKernelGGL((upfirdn_fir4_kernel<T, UP, DOWN, 0, 0>), g, b, 0, stream, a, tiles_x, tiles_y);
    else if (px == 0 && py == 0) hipLaunchKernelGGL((upfirdn_fir4_kernel<T, UP, DOWN, 0, 0>), g, b, 0, stream, a, tiles_x, tiles_y);
    else if (px == 1 && py == 0) hipLaunchKernelGGL((upfirdn_fir4_kernel<T, UP, DOWN, 1, 0>), g, b, 0, stream, a, tiles_x, tiles_y);
    else if (px == 0 && py == 1) hipLaunchKernelGGL((upfirdn_fir4_kernel<T, UP, DOWN, 0, 1>), g, b, 0, stream, a, tiles_x, tiles_y);
    else hipLaunchKernelGGL((upfirdn_fir4_kernel<T, UP, DOWN, 1, 1>), g, b, 0, stream, a, tiles_x, tiles_y);
    return check_launch("upfirdn2d(fir4)") == GNERF_OK ? 0 : -1;
}

template <class T>
int launch_up(const UpArgs& a, hipStream_t stream) {
    const bool nchw = a.xs_w == 1 && a.xs_h == a.in_w && a.xs_c == int64_t(a.in_h) * a.in_w && a.xs_n == a.xs_c * a.c &&
                      a.ys_w == 1 && a.ys_h == a.out_w && a.ys_c == int64_t(a.out_h) * a.out_w && a.ys_n == a.ys_c * a.c;
    const bool small = a.fh <= MAX_TAPS && a.fw <= MAX_TAPS && a.downx <= 2 && a.downy <= 2;
    if constexpr (sizeof(typename Arith<T>::type) == 4) {
        const bool fir4 = nchw && a.fh == 4 && a.fw == 4 && a.upx == a.upy && a.downx == a.downy &&
                          (reinterpret_cast<uintptr_t>(a.x) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.y) & 15) == 0;
        if (fir4) {
            int rc = 1;
            if (a.upx == 1 && a.downx == 1) rc = launch_blur4<T>(a, stream);
            else if (a.upx == 2 && a.downx == 1) rc = launch_fir4<T, 2, 1>(a, stream);
            else if (a.upx == 1 && a.downx == 2) rc = launch_fir4<T, 1, 2>(a, stream);
            if (rc == 0) return GNERF_OK;
            if (rc < 0) return GNERF_E_LAUNCH;
        }
    }
    if constexpr (sizeof(typename Arith<T>::type) == 4) {
        constexpr int VEC = 16 / sizeof(T);
        const bool nhwc = a.c > 1 && a.xs_c == 1 && a.xs_w == a.c && a.xs_h == int64_t(a.in_w) * a.c && a.xs_n == a.xs_h * a.in_h &&
                          a.ys_c == 1 && a.ys_w == a.c && a.ys_h == int64_t(a.out_w) * a.c && a.ys_n == a.ys_h * a.out_h;
        if (nhwc && a.fh == 4 && a.fw == 4 && a.upx == 1 && a.upy == 1 && a.downx == 1 && a.downy == 1 && a.c % VEC == 0 &&
            (reinterpret_cast<uintptr_t>(a.x) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.y) & 15) == 0) {
            const int rc = launch_blur4_nhwc<T>(a, stream);
            if (rc == 0) return GNERF_OK;
            if (rc < 0) return GNERF_E_LAUNCH;
        }
    }
    if (nchw && small && sizeof(typename Arith<T>::type) == 4) {      // LDS window is float: keep double on the direct kernel
        const int tiles_x = (a.out_w + TILE_W - 1) / TILE_W, tiles_y = (a.out_h + TILE_H - 1) / TILE_H;
        const int64_t blocks = int64_t(tiles_x) * tiles_y * a.n * a.c;
        if (blocks <= INT32_MAX) {
            hipLaunchKernelGGL((upfirdn_tile_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, stream, a, tiles_x, tiles_y);
            return check_launch("upfirdn2d(tile)");
        }
    }
    const int64_t total = int64_t(a.n) * a.c * a.out_h * a.out_w;
    int64_t blocks = (total + 255) / 256;
    if (blocks > int64_t(kNumCU) * 16) blocks = int64_t(kNumCU) * 16;
    const int channels_last = (a.ys_c == 1 && a.c > 1) ? 1 : 0;
    hipLaunchKernelGGL((upfirdn_any_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, stream, a, channels_last);
    return check_launch("upfirdn2d(any)");
}

}  // namespace

extern "C" int gnerf_upfirdn2d(const void* x, const float* f, void* y, int dtype,
                               int n, int c, int in_h, int in_w, const int64_t xs[4],
                               int fh, int fw, const int64_t fs[2],
                               int out_h, int out_w, const int64_t ys[4],
                               int upx, int upy, int downx, int downy, int padx0, int pady0,
                               int flip, float gain, gnerf_stream_t stream) {
    using namespace gnerf;
    if (!x || !f || !y || !xs || !fs || !ys) return fail(GNERF_E_ARG, "upfirdn2d: null pointer argument");
    if (n < 1 || c < 1 || in_h < 1 || in_w < 1) return fail(GNERF_E_ARG, "upfirdn2d: x has zero size");
    if (fh < 1 || fw < 1) return fail(GNERF_E_ARG, "upfirdn2d: f must be at least 1x1");
    if (upx < 1 || upy < 1) return fail(GNERF_E_ARG, "upfirdn2d: upsampling factor must be at least 1");
    if (downx < 1 || downy < 1) return fail(GNERF_E_ARG, "upfirdn2d: downsampling factor must be at least 1");
    if (out_h < 1 || out_w < 1) return fail(GNERF_E_ARG, "upfirdn2d: output must be at least 1x1");
    if (int64_t(n) * c * out_h * out_w > INT32_MAX || int64_t(n) * c * in_h * in_w > INT32_MAX)
        return fail(GNERF_E_ARG, "upfirdn2d: tensor is too large");
    UpArgs a{x, f, y, n, c, in_h, in_w, xs[0], xs[1], xs[2], xs[3], fh, fw, fs[0], fs[1], out_h, out_w,
             ys[0], ys[1], ys[2], ys[3], upx, upy, downx, downy, padx0, pady0, flip ? 1 : 0, gain};
    hipStream_t s = as_stream(stream);
    switch (dtype) {
        case GNERF_F32: return launch_up<float>(a, s);
        case GNERF_F16: return launch_up<__half>(a, s);
        case GNERF_F64: return launch_up<double>(a, s);
        default: return fail(GNERF_E_ARG, "upfirdn2d: unsupported dtype code %d", dtype);
    }
}

extern "C" int gnerf_blur4_epilogue_nhwc(const void* x, const float* f, void* y, int dtype, int n, int c, int in_h, int in_w, int out_h, int out_w,
                                         int padx0, int pady0, int flip, float blur_gain,
                                         const float* scale, const void* bias, int act, float alpha, float gain, float clamp, const float* next_scale,
                                         gnerf_stream_t stream) {
    using namespace gnerf;
    if (!x || !f || !y) return fail(GNERF_E_ARG, "blur4_epilogue_nhwc: null pointer argument");
    if (n < 1 || c < 1 || in_h < 1 || in_w < 1 || out_h < 1 || out_w < 1) return fail(GNERF_E_ARG, "blur4_epilogue_nhwc: empty shape");
    if (act != 1 && act != 3) return fail(GNERF_E_UNSUPPORTED, "blur4_epilogue_nhwc: only linear and lrelu");
    if (int64_t(n) * c * out_h * out_w > INT32_MAX || int64_t(n) * c * in_h * in_w > INT32_MAX) return fail(GNERF_E_ARG, "blur4_epilogue_nhwc: tensor is too large");
    const int es = dtype == GNERF_F16 ? 2 : (dtype == GNERF_F32 ? 4 : 0);
    if (!es) return fail(GNERF_E_ARG, "blur4_epilogue_nhwc: dtype must be float32 or float16");
    if (c % (16 / es) != 0 || ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(scale) |
                                reinterpret_cast<uintptr_t>(next_scale) | reinterpret_cast<uintptr_t>(bias)) & 15))
        return fail(GNERF_E_UNSUPPORTED, "blur4_epilogue_nhwc: channels must fill 16-byte vectors and the tensors be 16-byte aligned");
    UpArgs a{x, f, y, n, c, in_h, in_w, int64_t(in_h) * in_w * c, 1, int64_t(in_w) * c, c, 4, 4, 4, 1, out_h, out_w,
             int64_t(out_h) * out_w * c, 1, int64_t(out_w) * c, c, 1, 1, 1, 1, padx0, pady0, flip ? 1 : 0, blur_gain};
    const BlurEpi ep{scale, bias, next_scale, alpha, gain, clamp};
    const int rc = dtype == GNERF_F16 ? launch_blur4_nhwc<__half>(a, as_stream(stream), &ep, act) : launch_blur4_nhwc<float>(a, as_stream(stream), &ep, act);
    // rc < 0: launch_blur4_nhwc's check_launch() has already put "upfirdn2d(blur4, channels_last): <hip error>" into gnerf_last_error()
    return rc == 0 ? GNERF_OK : (rc < 0 ? GNERF_E_LAUNCH : fail(GNERF_E_ARG, "blur4_epilogue_nhwc: grid too large"));
}
