// Fused filtered leaky ReLU for gfx950: bias -> zero-insert upsample + FIR -> gain * lrelu * clamp (+2-bit
// signs) -> FIR + decimate, one launch, the 4x / 16x larger intermediate never leaves the CU.
//
// Replaces filtered_lrelu_plugin.filtered_lrelu (reference torch_utils/ops/filtered_lrelu.cpp:20-213,
// filtered_lrelu.cu:143-1103).  Same contract: returns GNERF_E_UNSUPPORTED when there is no kernel for the
// configuration, and the caller then takes the three-launch route (filtered_lrelu.py:225-231).  G-NeRF itself never
// executes this op; it is the StyleGAN3 layer primitive kept API-complete.
//
// Design (not the reference's: no constant-memory filter staging, no per-specialisation tile tables):
//   * one workgroup = one output tile of one (image, channel); all sizes are runtime values, only the resampling
//     factors (1/2/4), the taps per polyphase branch (1, 6 or 8) and the sign mode are compile-time, so the FIR inner loops
//     are fully unrolled with their taps in SGPRs.  Every workgroup builds the polyphase tap tables itself, from the filter
//     tensors into a few hundred bytes of LDS, and each pass reads its taps back with constant offsets (the reference's
//     filter set-up kernel + memcpy-to-constant and its stream hazard, filtered_lrelu.py:217-218, do not exist);
//   * four separable passes through two LDS buffers (fp32):  in -> (H up) -> (V up, activation, signs) -> (H down)
//     -> (V down) -> y.  A lane owns a line perpendicular to the filter axis (rows for the horizontal passes with an
//     odd row pitch, columns for the vertical ones), so every LDS access of a wave is bank-conflict free, and it
//     produces a block of outputs along the filter axis from one sliding register window: G*UP outputs from G+TU
//     reads (up), R outputs from (R-1)*DOWN + FD reads (down), with G = R = 2;
//   * the zero-insertion upsample is never materialised: output c of phase p reads input (c+k-pad)/up only for the
//     taps k = phase - p (mod up);
//   * signs: a lane's four neighbours in a row are the four pixels of one sign byte, combined with two DPP
//     quad permutes; a tile writes the bytes of the columns/rows it owns (its stride region), so overlapping
//     halos never write partial bytes.

#include "common.h"
#include <cstdlib>

namespace {

using namespace gnerf;

struct FlArgs {
    const void* x; void* y; const void* b; uint8_t* s; const float* fu; const float* fd;
    int dtype;
    int n, c, xh, xw, yh, yw;
    int64_t xs_n, xs_c, xs_h, xs_w, ys_n, ys_c, ys_h, ys_w;
    int fuw, fdw;               // taps (per axis)
    int fu_is2d, fd_is2d;       // 1x1 rank-2 filters apply once (x pass), not once per axis
    int px0, py0;
    int s_h, s_wb, sw_limit;    // sign tensor rows, bytes per row, bytes per row that are in use
    int sx, sy;
    float gain, slope, clamp;   // gain includes up*up
    int flip;
    int tow, toh, tiles_x, tiles_y;
    int pitch_in, pitch_up, pitch_dn;   // LDS row pitches (odd), sized so that no pass needs a bounds check
    int buf0_floats, buf1_floats;
};

constexpr int kMaxThreads = 512;        // the launcher picks 128, 256 or 512 lanes per workgroup (see gnerf_filtered_lrelu)
// Blocking along the filter axis, measured on the StyleGAN3-layer shapes (tools/bench_filtered_lrelu.py): 2 and 2 beat 1, 3, 4 and 8 --
// wider blocks reuse more of the register window but leave fewer work items than the 256 lanes want (4 / 4: 0.35 ms, 2 / 2: 0.30 ms at
// up 2 / down 2; 0.49 -> 0.40 ms at down 4).
#ifndef GNERF_FL_G
#define GNERF_FL_G 2
#endif
#ifndef GNERF_FL_R
#define GNERF_FL_R 2
#endif
constexpr int G = GNERF_FL_G;   // polyphase groups per work item (up passes)
constexpr int R = GNERF_FL_R;   // outputs per work item (down passes)

__host__ __device__ constexpr int round_up(int a, int b) { return (a + b - 1) / b * b; }

__device__ __forceinline__ int ceil_div_i(int a, int b) { return (a >= 0) ? (a + b - 1) / b : -((-a) / b); }

// i / d for 0 <= i < 2^20 with inv = 1.0f / d (exact: the half-step offset keeps the product away from integers)
__device__ __forceinline__ int fast_div(int i, float inv) { return int((float(i) + 0.5f) * inv); }

// f'[k]: tap k of the filter as correlated with the padded signal (flipped unless `flip`), zero outside.
__device__ __forceinline__ float tap(const float* __restrict__ f, int taps, int flip, int k) {
    const bool ok = (k >= 0) && (k < taps);
    const int kk = ok ? k : 0;
    const float v = f[flip ? kk : taps - 1 - kk];
    return ok ? v : 0.f;
}

__device__ __forceinline__ float uniform(float v) { return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(v))); }

// SIGN: 0 none, 1 write, 2 read
template <int UP, int TU, int DOWN, int TD, int SIGN>
__global__ __launch_bounds__(kMaxThreads) void filtered_lrelu_fused_kernel(FlArgs a) {
    extern __shared__ float lds[];
    constexpr int FD = TD * DOWN;
    constexpr int NFU = UP * (TU + 1);
    float* buf0 = lds;                                  // in, then V (activated intermediate)
    float* buf1 = lds + a.buf0_floats;                  // U1 (after H up), then D1 (after H down)
    float* taps = buf1 + a.buf1_floats;                 // [x up | y up | x down | y down] polyphase tables

    int bid = blockIdx.x;
    const int tx = bid % a.tiles_x; bid /= a.tiles_x;
    const int ty = bid % a.tiles_y; bid /= a.tiles_y;
    const int ch = bid % a.c;
    const int img = bid / a.c;
    const int tid = threadIdx.x;
    const int kThreads = blockDim.x;

    const int ox0 = tx * a.tow, oy0 = ty * a.toh;
    const int tow = min(a.tow, a.yw - ox0), toh = min(a.toh, a.yh - oy0);
    const int ux0 = ox0 * DOWN, uy0 = oy0 * DOWN;
    const int tuw = (tow - 1) * DOWN + FD, tuh = (toh - 1) * DOWN + FD;
    const int ngx = (tuw + UP - 1) / UP, ngy = (tuh + UP - 1) / UP;
    const int tiw = ngx + TU, tih = ngy + TU;
    const int ix0 = ceil_div_i(ux0 - a.px0, UP), iy0 = ceil_div_i(uy0 - a.py0, UP);
    const int phx = ix0 * UP - (ux0 - a.px0), phy = iy0 * UP - (uy0 - a.py0);
    const int PI = a.pitch_in, PU = a.pitch_up, PD = a.pitch_dn;

    // ---- polyphase tables -> LDS (one tap per lane; the passes read them back with constant offsets).
    //      up:   F[p][t] = f'[phase - p + t*UP], t = 0..TU;   down: F[k] = f'[k]
    for (int e0 = tid; e0 < 2 * NFU + 2 * FD; e0 += kThreads) {
        if (e0 < 2 * NFU) {
            const int yy = e0 >= NFU, e = e0 - yy * NFU;
            const int p = e / (TU + 1), t = e - p * (TU + 1);
            const int k = (yy ? phy : phx) - p + t * UP;
            float v = tap(a.fu, a.fuw, a.flip, k);
            if (yy && a.fu_is2d) v = (k == 0) ? 1.f : 0.f;
            taps[e0] = v;
        } else {
            const int e = e0 - 2 * NFU, yy = e >= FD, k = e - yy * FD;
            float v = tap(a.fd, a.fdw, a.flip, k);
            if (yy && a.fd_is2d) v = (k == 0) ? 1.f : 0.f;
            taps[e0] = v;
        }
    }

    // ---- S0: input tile + bias, zero outside the image.
    {
        const float bias = (a.dtype == GNERF_F16) ? __half2float(static_cast<const __half*>(a.b)[ch]) : static_cast<const float*>(a.b)[ch];
        const int64_t base = img * a.xs_n + ch * a.xs_c;
        const float inv = 1.0f / float(tiw);
        for (int i = tid; i < tih * tiw; i += kThreads) {
            const int r = fast_div(i, inv), q = i - r * tiw;
            const int gx = ix0 + q, gy = iy0 + r;
            float v = 0.f;
            if (unsigned(gx) < unsigned(a.xw) && unsigned(gy) < unsigned(a.xh)) {
                const int64_t o = base + gy * a.xs_h + gx * a.xs_w;
                v = ((a.dtype == GNERF_F16) ? __half2float(static_cast<const __half*>(a.x)[o]) : static_cast<const float*>(a.x)[o]) + bias;
            }
            buf0[r * PI + q] = v;
        }
    }
    __syncthreads();

    // ---- S1: horizontal upsampling FIR.  U1[r][g*UP+p] = sum_t in[r][g+t] * F[p][t].  A lane owns a row.
    {
        float F[UP][TU + 1];
#pragma unroll
        for (int p = 0; p < UP; p++)
#pragma unroll
            for (int t = 0; t <= TU; t++) F[p][t] = uniform(taps[p * (TU + 1) + t]);
        const int nbx = (ngx + G - 1) / G;
        const float inv = 1.0f / float(tih);
        for (int i = tid; i < nbx * tih; i += kThreads) {
            const int bx = fast_div(i, inv), r = i - bx * tih;
            const int g0 = bx * G;
            float w[G + TU];
#pragma unroll
            for (int k = 0; k < G + TU; k++) w[k] = buf0[r * PI + g0 + k];
#pragma unroll
            for (int gi = 0; gi < G; gi++)
#pragma unroll
                for (int p = 0; p < UP; p++) {
                    float acc = 0.f;
#pragma unroll
                    for (int t = 0; t <= TU; t++) acc = fmaf(w[gi + t], F[p][t], acc);
                    buf1[r * PU + (g0 + gi) * UP + p] = acc;
                }
        }
    }
    __syncthreads();

    // ---- S2: vertical upsampling FIR, gain, leaky ReLU, clamp, signs.  V[g*UP+p][x].  A lane owns a column.
    {
        float F[UP][TU + 1];
#pragma unroll
        for (int p = 0; p < UP; p++)
#pragma unroll
            for (int t = 0; t <= TU; t++) F[p][t] = uniform(taps[NFU + p * (TU + 1) + t]);
        const int tuw4 = (tuw + 3) & ~3;                       // lanes of a quad = the pixels of one sign byte
        const int nby = (ngy + G - 1) / G;
        const bool last_x = (tx == a.tiles_x - 1), last_y = (ty == a.tiles_y - 1);
        const int own_w = last_x ? tuw4 : a.tow * DOWN, own_h = min(last_y ? tuh : a.toh * DOWN, tuh);
        const int64_t splane = (int64_t(img) * a.c + ch) * a.s_h;
        const float inv = 1.0f / float(tuw4);
        for (int i = tid; i < nby * tuw4; i += kThreads) {
            const int by = fast_div(i, inv), x = i - by * tuw4;
            const int g0 = by * G;
            float w[G + TU];
#pragma unroll
            for (int k = 0; k < G + TU; k++) w[k] = buf1[(g0 + k) * PU + x];
            const int sgx = ux0 + x + a.sx;
            const bool sx_ok = unsigned(sgx >> 2) < unsigned(a.sw_limit);
            const int sshift = (sgx & 3) << 1;
#pragma unroll
            for (int gi = 0; gi < G; gi++)
#pragma unroll
                for (int p = 0; p < UP; p++) {
                    float acc = 0.f;
#pragma unroll
                    for (int t = 0; t <= TU; t++) acc = fmaf(w[gi + t], F[p][t], acc);
                    const int j = (g0 + gi) * UP + p;
                    float v = acc * a.gain;
                    const int sgy = uy0 + j + a.sy;
                    if (SIGN == 2) {
                        if (sx_ok && unsigned(sgy) < unsigned(a.s_h) && j < tuh) {
                            const unsigned sb = unsigned(a.s[(splane + sgy) * a.s_wb + (sgx >> 2)]) >> sshift;
                            if (sb & 1) v *= a.slope;
                            if (sb & 2) v = 0.f;
                        }
                    } else {
                        int sg = int(__float_as_uint(v) >> 31);
                        v = sg ? v * a.slope : v;
                        const bool cl = fabsf(v) > a.clamp;
                        v = cl ? copysignf(a.clamp, v) : v;
                        if (SIGN == 1) {
                            sg = cl ? 2 : sg;
                            sg = (x < tuw) ? sg : 0;
                            int bits = sg << sshift;
                            bits |= __builtin_amdgcn_mov_dpp(bits, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
                            bits |= __builtin_amdgcn_mov_dpp(bits, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
                            if ((x & 3) == 0 && x < own_w && j < own_h && sx_ok && unsigned(sgy) < unsigned(a.s_h))
                                a.s[(splane + sgy) * a.s_wb + (sgx >> 2)] = uint8_t(bits);
                        }
                    }
                    buf0[j * PU + x] = v;
                }
        }
    }
    __syncthreads();

    // ---- S3: horizontal downsampling FIR.  D1[r][o] = sum_k V[r][o*DOWN + k] * F[k].  A lane owns a row.
    {
        float F[FD];
#pragma unroll
        for (int k = 0; k < FD; k++) F[k] = uniform(taps[2 * NFU + k]);
        const int nbo = (tow + R - 1) / R;
        const float inv = 1.0f / float(tuh);
        for (int i = tid; i < nbo * tuh; i += kThreads) {
            const int bo = fast_div(i, inv), r = i - bo * tuh;
            const int o0 = bo * R;
            float w[(R - 1) * DOWN + FD];
#pragma unroll
            for (int k = 0; k < (R - 1) * DOWN + FD; k++) w[k] = buf0[r * PU + o0 * DOWN + k];
#pragma unroll
            for (int ri = 0; ri < R; ri++) {
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < FD; k++) acc = fmaf(w[ri * DOWN + k], F[k], acc);
                buf1[r * PD + o0 + ri] = acc;
            }
        }
    }
    __syncthreads();

    // ---- S4: vertical downsampling FIR and the store.  A lane owns a column.
    {
        float F[FD];
#pragma unroll
        for (int k = 0; k < FD; k++) F[k] = uniform(taps[2 * NFU + FD + k]);
        const int nbo = (toh + R - 1) / R;
        const int64_t base = img * a.ys_n + ch * a.ys_c;
        const float inv = 1.0f / float(tow);
        for (int i = tid; i < nbo * tow; i += kThreads) {
            const int bo = fast_div(i, inv), x = i - bo * tow;
            const int o0 = bo * R;
            float w[(R - 1) * DOWN + FD];
#pragma unroll
            for (int k = 0; k < (R - 1) * DOWN + FD; k++) w[k] = buf1[(o0 * DOWN + k) * PD + x];
#pragma unroll
            for (int ri = 0; ri < R; ri++) {
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < FD; k++) acc = fmaf(w[ri * DOWN + k], F[k], acc);
                if (o0 + ri < toh) {
                    const int64_t o = base + (oy0 + o0 + ri) * a.ys_h + (ox0 + x) * a.ys_w;
                    if (a.dtype == GNERF_F16) static_cast<__half*>(a.y)[o] = __float2half(acc);
                    else static_cast<float*>(a.y)[o] = acc;
                }
            }
        }
    }
}

// LDS geometry for a nominal tile: every pass reads and writes whole G- or R-blocks without bounds checks, so the
// buffers are padded to block multiples (out-of-tile entries hold don't-care values that only feed don't-care outputs).
struct FlGeom { int pitch_in, pitch_up, pitch_dn, buf0, buf1, taps; size_t bytes; };

FlGeom geometry(int up, int tu, int down, int td, int tow, int toh) {
    const int fd = td * down;
    const int tuw = (tow - 1) * down + fd, tuh = (toh - 1) * down + fd;
    const int ngx = (tuw + up - 1) / up, ngy = (tuh + up - 1) / up;
    const int tih = ngy + tu;
    FlGeom g;
    g.pitch_in = (round_up(ngx, G) + tu) | 1;                                              // S1 reads in[r][g0 .. g0+G+TU)
    g.pitch_up = std::max(std::max(round_up(ngx, G) * up, round_up(tuw, 4)), (round_up(tow, R) - 1) * down + fd) | 1;
    g.pitch_dn = round_up(tow, R) | 1;
    const int u1_rows = std::max(tih, round_up(ngy, G) + tu);                              // S2 reads U1[g0 .. g0+G+TU)
    const int v_rows = std::max(round_up(ngy, G) * up, (round_up(toh, R) - 1) * down + fd);  // S2 writes, S4 reads through D1
    g.buf0 = std::max(tih * g.pitch_in, v_rows * g.pitch_up);
    g.buf1 = std::max(u1_rows * g.pitch_up, v_rows * g.pitch_dn);
    g.taps = 2 * up * (tu + 1) + 2 * fd;
    g.bytes = size_t(g.buf0 + g.buf1 + g.taps) * sizeof(float);
    return g;
}

typedef void (*FlKernel)(FlArgs);

template <int UP, int TU, int DOWN, int TD>
FlKernel pick_sign(int sign) {
    if (sign == 0) return filtered_lrelu_fused_kernel<UP, TU, DOWN, TD, 0>;
    if (sign == 1) return filtered_lrelu_fused_kernel<UP, TU, DOWN, TD, 1>;
    return filtered_lrelu_fused_kernel<UP, TU, DOWN, TD, 2>;
}

template <int UP, int TU>
FlKernel pick_down(int down, int td, int sign) {
    if (down == 1 && td == 1) return pick_sign<UP, TU, 1, 1>(sign);
    if (down == 2 && td == 6) return pick_sign<UP, TU, 2, 6>(sign);
    if (down == 2 && td == 8) return pick_sign<UP, TU, 2, 8>(sign);
    if (down == 4 && td == 6) return pick_sign<UP, TU, 4, 6>(sign);
    if (down == 4 && td == 8) return pick_sign<UP, TU, 4, 8>(sign);
    return nullptr;
}

FlKernel pick(int up, int tu, int down, int td, int sign) {
    if (up == 1 && tu == 1) return pick_down<1, 1>(down, td, sign);
    if (up == 2 && tu == 6) return pick_down<2, 6>(down, td, sign);
    if (up == 2 && tu == 8) return pick_down<2, 8>(down, td, sign);
    if (up == 4 && tu == 6) return pick_down<4, 6>(down, td, sign);
    if (up == 4 && tu == 8) return pick_down<4, 8>(down, td, sign);
    return nullptr;
}

// taps per polyphase branch the kernels are built for: 1 (no resampling, one tap), 6, 8
int branch_taps(int factor, int taps) {
    if (factor == 1) return taps == 1 ? 1 : 0;
    if (taps < factor) return 0;
    if (taps <= 6 * factor) return 6;
    if (taps <= 8 * factor) return 8;
    return 0;
}

}  // namespace

extern "C" int gnerf_filtered_lrelu(const void* x, const float* fu, const float* fd, const void* b, uint8_t* s, void* y,
                                    int dtype, int n, int c, int xh, int xw, const int64_t xs[4],
                                    int yh, int yw, const int64_t ys[4],
                                    int fu_taps, int fu_rank, int fd_taps, int fd_rank,
                                    int up, int down, int px0, int py0,
                                    int s_h, int s_w, int sx, int sy, int sign_mode,
                                    float gain, float slope, float clamp, int flip, gnerf_stream_t stream) {
    using namespace gnerf;
    if (!x || !fu || !fd || !b || !y || !xs || !ys) return fail(GNERF_E_ARG, "filtered_lrelu: null pointer");
    if (n < 1 || c < 1 || xh < 1 || xw < 1 || yh < 1 || yw < 1) return fail(GNERF_E_ARG, "filtered_lrelu: empty tensor");
    if (up < 1 || down < 1) return fail(GNERF_E_ARG, "filtered_lrelu: up and down must be at least 1");
    if (fu_taps < 1 || fd_taps < 1) return fail(GNERF_E_ARG, "filtered_lrelu: empty filter");
    if (sign_mode < 0 || sign_mode > 2) return fail(GNERF_E_ARG, "filtered_lrelu: sign_mode must be 0, 1 or 2");
    if (sign_mode != 0 && (!s || s_h < 1 || s_w < 4 || (s_w & 3))) return fail(GNERF_E_ARG, "filtered_lrelu: bad sign tensor");
    if (dtype != GNERF_F32 && dtype != GNERF_F16) return fail(GNERF_E_UNSUPPORTED, "filtered_lrelu: float16/float32 only");
    // rank-2 filters: only 1x1 (what the Python layer builds for "no filter"); real 2-D filters take the generic route
    if ((fu_rank == 2 && fu_taps != 1) || (fd_rank == 2 && fd_taps != 1)) return fail(GNERF_E_UNSUPPORTED, "filtered_lrelu: no fused kernel for non-separable filters");
    const int tu = branch_taps(up, fu_taps), td = branch_taps(down, fd_taps);
    FlKernel k = (tu && td) ? pick(up, tu, down, td, sign_mode) : nullptr;
    if (!k) return fail(GNERF_E_UNSUPPORTED, "filtered_lrelu: no fused kernel for up=%d (%d taps) down=%d (%d taps)", up, fu_taps, down, fd_taps);
    if (sign_mode == 1 && ((sx & 3) || sy < 0 || sx < 0)) return fail(GNERF_E_UNSUPPORTED, "filtered_lrelu: sign write needs a 4-aligned sign offset");
    if (int64_t(n) * c > (1 << 30)) return fail(GNERF_E_UNSUPPORTED, "filtered_lrelu: too many feature maps");

    FlArgs a{};
    a.x = x; a.y = y; a.b = b; a.s = s; a.fu = fu; a.fd = fd; a.dtype = dtype;
    a.n = n; a.c = c; a.xh = xh; a.xw = xw; a.yh = yh; a.yw = yw;
    a.xs_n = xs[0]; a.xs_c = xs[1]; a.xs_h = xs[2]; a.xs_w = xs[3];
    a.ys_n = ys[0]; a.ys_c = ys[1]; a.ys_h = ys[2]; a.ys_w = ys[3];
    a.fuw = fu_taps; a.fdw = fd_taps; a.fu_is2d = (fu_rank == 2); a.fd_is2d = (fd_rank == 2);
    a.px0 = px0; a.py0 = py0;
    a.s_h = s_h; a.s_wb = s_w >> 2; a.sx = sx; a.sy = sy;
    // bytes per row in use: when writing, the active width (filtered_lrelu.cpp:96,137); when reading, the whole row
    a.sw_limit = (sign_mode == 1) ? int(((int64_t(yw) * down - (down - 1) + (fd_taps - 1)) + 3) >> 2) : (s_w >> 2);
    a.gain = float(up) * float(up) * gain; a.slope = slope; a.clamp = clamp; a.flip = flip ? 1 : 0;

    // Largest tile whose LDS buffers fit the budget; tile width * down stays a multiple of 4 (sign bytes).
    static const int cand[][2] = {{32, 32}, {32, 16}, {16, 16}, {16, 8}, {8, 8}, {8, 4}, {4, 4}};
    const size_t budget = 48 * 1024;
    size_t bytes = 0;
    bool found = false;
    int forced[2] = {0, 0};
    if (const char* e = getenv("GNERF_FL_TILE")) sscanf(e, "%d,%d", &forced[0], &forced[1]);        // A/B aid: force a tile (width*down must stay a multiple of 4)
    if (forced[0] > 0 && forced[1] > 0 && (forced[0] * down) % 4 == 0) {
        const FlGeom g = geometry(up, tu, down, td, forced[0], forced[1]);
        if (g.bytes <= 64 * 1024) {
            a.tow = forced[0]; a.toh = forced[1];
            a.pitch_in = g.pitch_in; a.pitch_up = g.pitch_up; a.pitch_dn = g.pitch_dn; a.buf0_floats = g.buf0; a.buf1_floats = g.buf1;
            bytes = g.bytes;
            found = true;
        }
    }
    for (const auto& t : cand) {
        if (found) break;
        const FlGeom g = geometry(up, tu, down, td, t[0], t[1]);
        if (g.bytes <= budget) {
            a.tow = t[0]; a.toh = t[1];
            a.pitch_in = g.pitch_in; a.pitch_up = g.pitch_up; a.pitch_dn = g.pitch_dn; a.buf0_floats = g.buf0; a.buf1_floats = g.buf1;
            bytes = g.bytes;
            found = true;
            break;
        }
    }
    if (!found) return fail(GNERF_E_UNSUPPORTED, "filtered_lrelu: tile does not fit in LDS");
    a.tiles_x = (yw + a.tow - 1) / a.tow;
    a.tiles_y = (yh + a.toh - 1) / a.toh;
    const int64_t blocks = int64_t(a.tiles_x) * a.tiles_y * n * c;
    if (blocks > 0x7fffffffLL) return fail(GNERF_E_UNSUPPORTED, "filtered_lrelu: grid too large");
    // Lanes per workgroup, measured per resampling pair (tools/bench_filtered_lrelu.py): the passes of a tile offer between ~250 (no
    // downsampling filter) and ~2 000 (down 4) work items each
    int threads = down == 1 ? 128 : (down == 4 ? 512 : 256);
    if (const char* e = getenv("GNERF_FL_THREADS")) { const int t = atoi(e); if (t == 128 || t == 256 || t == 512) threads = t; }       // A/B aid
    hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(threads), bytes, as_stream(stream), a);
    return check_launch("filtered_lrelu");
}
