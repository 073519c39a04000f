// Fused 3x3 convolution + modulated-convolution epilogue for the superresolution's fp16 channels_last layers (SURVEY.md section 8(f)3:
// the convolution inside modulated_conv2d, networks_stylegan2.py:41-98, as SynthesisLayer.forward calls it, :315-334, for
// superresolution.py:285-303's blocks).
//
// In gnerf_generator's shared-weight form (one latent per batch) a layer is
//     x' = x * styles[n, c]            (folded into the previous layer's epilogue)
//     y  = conv3x3(x', w)              <- MIOpen / CK until round 4: 0.31-0.35 of the f16 matrix peak on these shapes
//     y  = clamp(lrelu(y * dcoefs[n, o] + noise + bias[o]) * gain) * next_styles[n, o]      <- gnerf_modconv_epilogue_nhwc
// This kernel does the last two lines in one launch: an implicit GEMM on v_mfma_f32_16x16x32_f16 whose accumulators go through the
// epilogue in registers (the same device function as the stand-alone pass, csrc/common.h: same roundings, in the same order) and
// leave as fp16 channels_last.
//
// Decomposition.  PERSISTENT workgroups (4 waves, 1 per SIMD, one workgroup per CU: 154 KB of LDS) walk a sequence of jobs; a job is an
// 8 x 32 tile of output pixels for 128 output channels.  A job is 9 taps x (Cin / 64) chunks = STEPS of 64 MFMAs per wave:
//   * the 10 x 34 input pixels a tile's nine taps touch are staged once per 64 input channels (43 KB; image borders come in as zeros
//     from the buffer load's range check), DOUBLE buffered: the next chunk -- of this job or of the next -- streams in while the nine
//     taps of the current one are multiplied, every tap reading the tile at a shifted position;
//   * the weights of a step (128 x 64, 16 KB) go through a ring of four buffers, requested three steps ahead;
//   * both by LDS-DMA (buffer_load / global_load ... lds: no registers, no ds_write), whose LDS image is lane-linear, so the
//     bank-conflict swizzle (16-byte slot ^ (row & 7) in 128-byte rows) is applied to the SOURCE address and again on the fragment reads;
//   * ONE workgroup barrier per step; the fragments of a step's first k-step are read during the previous step's last MFMAs (the data
//     they need was waited for and published by the barrier before), so the matrix pipe sees a continuous instruction stream across
//     taps, chunks and jobs.  Only the epilogue of a job is not overlapped.
// (The first version -- one workgroup per tile -- spent as long between tiles as in them: at one workgroup per CU a new workgroup starts
//  only when the old one has released its LDS, then loads its tile, then computes: 52 k cycles per tile for 18 k cycles of MFMAs.)
// Orientation: A = weights (M = 16 output channels), B = input (N = 16 pixels of a row), K = 32 input channels per instruction; a lane's
// four accumulator registers are then four CONSECUTIVE output channels of one pixel -- an 8-byte piece of the channels_last result.
// A wave owns two rows of the tile: 64 pixels x 128 channels = 128 accumulator registers, 32 MFMAs per 12 ds_read_b128.

#include "common.h"

namespace {

using namespace gnerf;

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int kTH = 8, kTW = 32;                    // output pixels of a tile
constexpr int kIH = kTH + 2, kIW = kTW + 2;         // input pixels incl. the one-pixel halo
constexpr int kCK = 64;                             // input channels of a chunk (one input-tile buffer)
constexpr int kCO = 128;                            // output channels per job
constexpr int kConvThreads = 256;
constexpr int kRow = kCK * 2;                       // bytes of a pixel's / an output channel's row in LDS (8 slots of 16 bytes)
constexpr int kXPieces = kIH * kIW * (kCK / 8);     // 16-byte pieces of an input tile
constexpr int kXRounds = (kXPieces + kConvThreads - 1) / kConvThreads;
constexpr int kXBytes = kXRounds * kConvThreads * 16;
constexpr int kWBytes = kCO * kRow;
constexpr int kWRounds = kWBytes / 16 / kConvThreads;
constexpr int kConvLds = 2 * kXBytes + 4 * kWBytes + 1536;       // + the epilogue's per-channel operands
static_assert(kTH * kTW * (kCO / 2) * 2 <= kXBytes, "half of the output tile (64 channels) is staged in one input-tile buffer");
static_assert(kCK == 64, "two k-steps per step; the swizzles assume 128-byte rows");

struct ConvArgs {
    const _Float16* x;          // [n, h, w, cin]    channels_last activations
    const _Float16* wpk;        // [9, cout, cin]     tap-major weights: tap = ky * 3 + kx of the correlation form
    _Float16* y;                // [n, h, w, cout]
    const float* scale;         // [n, cout] or NULL  (demodulation coefficients)
    const float* noise;         // [h * w] or NULL
    const __half* bias;         // [cout] or NULL
    const float* next_scale;    // [n, cout] or NULL
    int n, h, w, cin, cout;
    int tiles_x, tiles_y, n_jobs, groups;
    int round_noise;
    float alpha, gain, clamp;
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;

// a step of a workgroup's sequence: tap `tap` of chunk `chunk` of its k-th job
struct ConvPos { int k, chunk, tap; };

template <bool SCALE, bool NOISE, bool NEXT>
__global__ __launch_bounds__(kConvThreads, 1) void conv3x3_epilogue_kernel(ConvArgs a) {
    extern __shared__ __align__(16) char lds[];
    char* const xb = lds;                           // 2 x [340 pixels][8 slots of 16 bytes], slot ^= pixel & 7
    char* const wb = lds + 2 * kXBytes;             // 4 x [128 output channels][8 slots], slot ^= channel & 7
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, hq = lane >> 4;
    const int H = a.h, W = a.w, Cin = a.cin, Cout = a.cout;
    const int n_chunks = Cin / kCK;

    // Jobs of this workgroup.  Workgroups b, b + 8, ... share an XCD (round-robin dispatch) and the XCD owns a contiguous eighth of the
    // job sequence, dealt to its workgroups one job at a time: at any moment they work on neighbouring tiles, whose halo pixels -- and
    // the weights -- the XCD's L2 then holds.  (Speed only: any assignment is correct.)
    const int per_xcd = (a.n_jobs + kNumXCD - 1) / kNumXCD;
    const int xcd = blockIdx.x % kNumXCD, local = blockIdx.x / kNumXCD, n_local = gridDim.x / kNumXCD;
    const int j_lo = xcd * per_xcd, j_hi = min(a.n_jobs, j_lo + per_xcd);
    auto job_of = [&](int k) { const int j = j_lo + local + k * n_local; return j < j_hi ? j : -1; };
    if (job_of(0) < 0) return;
    struct Tile { int n, y0, x0, co0; };
    auto tile_of = [&](int job) {
        const int t = job / a.groups, cg = job - t * a.groups;
        const int tiles_img = a.tiles_x * a.tiles_y;
        const int n = t / tiles_img, tt = t - n * tiles_img;
        const int ty = tt / a.tiles_x, tx = tt - ty * a.tiles_x;
        return Tile{n, ty * kTH, tx * kTW, cg * kCO};
    };
    auto advance = [&](ConvPos p) {
        if (++p.tap == 9) { p.tap = 0; if (++p.chunk == n_chunks) { p.chunk = 0; p.k++; } }
        return p;
    };
    auto valid = [&](const ConvPos& p) { return job_of(p.k) >= 0; };

    auto stage_x = [&](const ConvPos& p, int buf) {               // the input tile of chunk p.chunk of job p.k
        const Tile t = tile_of(job_of(p.k));
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<_Float16*>(a.x) + size_t(t.n) * H * W * Cin, 0, int(size_t(H) * W * Cin * 2), 0x00020000);
        const int cin0 = p.chunk * kCK;
#pragma unroll
        for (int it = 0; it < kXRounds; it++) {
            const int q = it * kConvThreads + tid;
            const int pix = q >> 3, slot = q & 7;
            const int py = pix / kIW, px = pix - py * kIW;
            const int iy = t.y0 - 1 + py, ix = t.x0 - 1 + px;
            const bool ok = pix < kIH * kIW && iy >= 0 && iy < H && ix >= 0 && ix < W;
            const unsigned off = ok ? unsigned(((iy * W + ix) * Cin + cin0 + ((slot ^ (pix & 7)) << 3)) * 2) : 0x80000000u;     // out of range: zeros
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(xb + buf * kXBytes + (it * kConvThreads + wv * 64) * 16), 16, off, 0, 0, 0);
        }
    };
    auto stage_w = [&](const ConvPos& p, int buf) {               // the weights of step p
        const Tile t = tile_of(job_of(p.k));
        const _Float16* src = a.wpk + (size_t(p.tap) * Cout + t.co0) * Cin + p.chunk * kCK;
#pragma unroll
        for (int it = 0; it < kWRounds; it++) {
            const int q = it * kConvThreads + tid;
            const int co = q >> 3, slot = q & 7;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + size_t(co) * Cin + ((slot ^ (co & 7)) << 3)),
                                             (lds_ptr_t)(wb + buf * kWBytes + (it * kConvThreads + wv * 64) * 16), 16, 0, 0);
        }
    };

    // fragment f of k-step kc of a step: f = 0..3 the input fragments of the wave's four pixel blocks, f = 4..11 the eight weight fragments
    struct Frags { h8 A[8], B[4]; };
    auto load_frag = [&](Frags& F, int f, int kc, const char* xbuf, const char* wbuf, int tap) {
        const int ks = kc * 4 + hq;
        if (f < 4) {
            const int dy = tap / 3, dx = tap - dy * 3;
            const int pi = (2 * wv + (f >> 1) + dy) * kIW + (f & 1) * 16 + r + dx;
            F.B[f] = *reinterpret_cast<const h8*>(xbuf + pi * kRow + ((ks ^ (pi & 7)) << 4));
        } else {
            F.A[f - 4] = *reinterpret_cast<const h8*>(wbuf + ((f - 4) * 16 + r) * kRow + ((ks ^ (r & 7)) << 4));
        }
    };

    v4f acc[8][4];
    auto zero_acc = [&]() {
#pragma unroll
        for (int cb = 0; cb < 8; cb++)
#pragma unroll
            for (int pb = 0; pb < 4; pb++) acc[cb][pb] = (v4f){0.f, 0.f, 0.f, 0.f};
    };
    auto mfma_group = [&](const Frags& F, int g) {                 // the MFMAs of output-channel blocks 2g, 2g + 1
#pragma unroll
        for (int cb = 2 * g; cb < 2 * g + 2; cb++)
#pragma unroll
            for (int pb = 0; pb < 4; pb++) acc[cb][pb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(F.A[cb], F.B[pb], acc[cb][pb], 0, 0, 0);
    };

    // ---- the epilogue of a job: registers -> (through half an LDS buffer per 64 channels, for 16-byte coalesced stores) -> y.
    // Its per-channel operands (demodulation scale, next layer's scale, bias of the job's 128 channels: 1.25 KB) come in by LDS-DMA at
    // the TOP of the job's last step and land under its MFMAs.  (As ordinary loads at the point of use they were 24 dependent round
    // trips per job; as ordinary loads issued early they made hipcc put `s_waitcnt vmcnt(0)` between the LDS-DMA instructions of the
    // input-tile prefetch -- its wait insertion does not count the two kinds of load apart.  So: LDS-DMA only.)
    char* const ep = lds + 2 * kXBytes + 4 * kWBytes;              // [128] float scale, [128] float next_scale, [128] half bias
    auto fetch_operands = [&](const Tile& t) {
        if (wv < 2) {                                              // (wave-uniform: lanes 0..127 of the workgroup, one float each)
            if constexpr (SCALE) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.scale + size_t(t.n) * Cout + t.co0 + tid),
                                                                  (lds_ptr_t)(ep + wv * 256), 4, 0, 0);
            if constexpr (NEXT) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.next_scale + size_t(t.n) * Cout + t.co0 + tid),
                                                                 (lds_ptr_t)(ep + 512 + wv * 256), 4, 0, 0);
        }
        if (wv == 0 && a.bias) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.bias + t.co0 + 2 * tid),
                                                                (lds_ptr_t)(ep + 1024), 4, 0, 0);
    };
    auto epilogue = [&](const Tile& t, char* os) {
#pragma unroll
        for (int hh = 0; hh < 2; hh++) {
#pragma unroll
            for (int cbh = 0; cbh < 4; cbh++) {
                const int cb = 4 * hh + cbh;
                const int c4 = cb * 16 + hq * 4;
                float sc[4] = {1.f, 1.f, 1.f, 1.f}, nx[4] = {1.f, 1.f, 1.f, 1.f}, bv[4] = {0.f, 0.f, 0.f, 0.f};
                if constexpr (SCALE) { const float4 v = *reinterpret_cast<const float4*>(ep + c4 * 4); sc[0] = v.x; sc[1] = v.y; sc[2] = v.z; sc[3] = v.w; }
                if constexpr (NEXT) {
                    const float4 v = *reinterpret_cast<const float4*>(ep + 512 + c4 * 4);
                    nx[0] = round_to<__half>(v.x); nx[1] = round_to<__half>(v.y); nx[2] = round_to<__half>(v.z); nx[3] = round_to<__half>(v.w);
                }
                if (a.bias) {
                    const uint2 hb2 = *reinterpret_cast<const uint2*>(ep + 1024 + c4 * 2);
                    const __half* hb = reinterpret_cast<const __half*>(&hb2);
#pragma unroll
                    for (int k = 0; k < 4; k++) bv[k] = __half2float(hb[k]);
                }
                const int c4h = c4 - 64 * hh;                      // within this half
#pragma unroll
                for (int pb = 0; pb < 4; pb++) {
                    const int prow = 2 * wv + (pb >> 1), pcol = (pb & 1) * 16 + r;
                    const int p = prow * kTW + pcol;
                    Pk<__half, 4> in;
#pragma unroll
                    for (int k = 0; k < 4; k++) in.v[k] = __float2half(acc[cb][pb][k]);          // what the convolution alone would have stored
                    float nz = 0.f;
                    if constexpr (NOISE) nz = a.noise[(t.y0 + prow) * W + t.x0 + pcol];
                    const Pk<__half, 4> out = modconv_epilogue_vec<__half, 4, 3, SCALE, NOISE, NEXT>(in, sc, nz, a.round_noise != 0, bv, nx, a.alpha, a.gain, a.clamp);
                    *reinterpret_cast<uint2*>(os + p * kRow + (((c4h >> 3) ^ (p & 7)) << 4) + ((c4h >> 2) & 1) * 8) = *reinterpret_cast<const uint2*>(&out);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (bare barriers: the next job's weights are in flight, see the main loop)
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
#pragma unroll
            for (int it = 0; it < kTH * kTW * 8 / kConvThreads; it++) {
                const int q = it * kConvThreads + tid;
                const int p = q >> 3, slot = q & 7;
                const uint4 v = *reinterpret_cast<const uint4*>(os + p * kRow + ((slot ^ (p & 7)) << 4));
                const int yy = t.y0 + p / kTW, xx = t.x0 + (p & (kTW - 1));
                *reinterpret_cast<uint4*>(a.y + (size_t(t.n) * H * W + size_t(yy) * W + xx) * Cout + t.co0 + 64 * hh + slot * 8) = v;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                          // the buffer is free again (for the other half, or for the next input tile)
            asm volatile("" ::: "memory");
        }
    };

    // ---- the pipeline.  Invariant at the top of step s (position `cur`, weights in ring slot `ring`, input tile in xb[xi]):
    //   W(s), W(s+1) and the input tile of step s+1 have landed and are visible to every wave; W(s+2) is in flight; the fragments of
    //   step s's first k-step are in F0; ring slot ring + 3 and xb[xi ^ 1] (at a chunk's first tap) are free.
    // Weights are requested THREE steps (3 x 64 MFMAs = 3 k cycles) before their first fragment read: under this kernel's own load a
    // global -> LDS transfer takes ~3 k cycles to land (with two steps' notice every step ended in a wait for it).
    ConvPos cur{0, 0, 0};
    ConvPos n1 = advance(cur), n2 = advance(n1), n3 = advance(n2);
    stage_x(cur, 0);
    stage_w(cur, 0);
    if (valid(n1)) stage_w(n1, 1);
    if (valid(n2)) stage_w(n2, 2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    Frags F0, F1;
#pragma unroll
    for (int f = 0; f < 12; f++) load_frag(F0, f, 0, xb, wb, 0);
    zero_acc();
    int xi = 0, ring = 0;                                           // input-tile buffer and ring slot of the current step
    for (;;) {
        const bool more1 = valid(n1);
        const bool last_step = cur.tap == 8 && cur.chunk == n_chunks - 1;
        if (last_step) fetch_operands(tile_of(job_of(cur.k)));
        // loads, oldest first: (the epilogue's operands,) the weights of step s+3, and at a chunk's first tap the input tile of the NEXT
        // chunk (of this job or of the next one)
        const bool w_issued = valid(n3);
        if (w_issued) stage_w(n3, (ring + 3) & 3);
        bool x_issued = false;
        if (cur.tap == 0) {
            ConvPos nc = cur;
            if (++nc.chunk == n_chunks) { nc.chunk = 0; nc.k++; }
            if (valid(nc)) { stage_x(nc, xi ^ 1); x_issued = true; }
        }
        const char* xcur = xb + xi * kXBytes;
        const char* wcur = wb + ring * kWBytes;
        // where step s+1 reads: the next ring slot; the other input-tile buffer when s+1 opens a chunk
        const int ring1 = (ring + 1) & 3;
        const int xi1 = n1.tap == 0 ? xi ^ 1 : xi;
        const char* xnext = xb + xi1 * kXBytes;
        const char* wnext = wb + ring1 * kWBytes;
        // ---- k-step 0 (fragments in F0), reading k-step 1's fragments into F1 meanwhile
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(F0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int f = 0; f < 6; f++) load_frag(F1, f, 1, xcur, wcur, cur.tap);
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(F0, 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int f = 6; f < 12; f++) load_frag(F1, f, 1, xcur, wcur, cur.tap);
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(F0, 2);
        mfma_group(F0, 3);
        // ---- k-step 1 (fragments in F1), reading the NEXT step's first fragments into F0 meanwhile
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(F1, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (more1) {
#pragma unroll
            for (int f = 0; f < 6; f++) load_frag(F0, f, 0, xnext, wnext, n1.tap);
        }
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(F1, 1);
        __builtin_amdgcn_sched_barrier(0);
        if (more1) {
#pragma unroll
            for (int f = 6; f < 12; f++) load_frag(F0, f, 0, xnext, wnext, n1.tap);
        }
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(F1, 2);
        mfma_group(F1, 3);
        __builtin_amdgcn_sched_barrier(0);
        // ---- end of the step: W(s+2) -- requested a step ago -- must have landed before the barrier publishes it; what this step requested
        // (W(s+3), then possibly an input tile: loads complete in order) stays in flight
        if (w_issued && x_issued)  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(kWRounds + kXRounds) : "memory");
        else if (x_issued)         asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(kXRounds) : "memory");
        else if (w_issued)         asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(kWRounds) : "memory");
        else                       asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        // (a bare s_barrier: __syncthreads() carries a fence that hipcc turns into vmcnt(0) while an LDS-DMA is in flight, which would
        //  drain the prefetches at every step; the waits this hand-off needs are the ones above)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (last_step) {                                            // the job's accumulators leave
            epilogue(tile_of(job_of(cur.k)), xb + xi * kXBytes);    // (this job's last input tile is dead; the other buffer holds the next job's first)
            zero_acc();
        }
        if (!more1) break;
        xi = xi1; ring = ring1;
        cur = n1; n1 = n2; n2 = n3; n3 = advance(n3);
    }
}

}  // namespace

// y = epilogue(conv3x3(x, w)) on float16 channels_last tensors; see include/gnerf_hip.h.
extern "C" int gnerf_conv3x3_epilogue_nhwc(const void* x, const void* w_packed, void* y, int n, int h, int w, int cin, int cout,
                                           const float* scale, const float* noise, int round_noise, const void* bias,
                                           float alpha, float gain, float clamp, const float* next_scale, gnerf_stream_t stream) {
    using namespace gnerf;
    if (!x || !w_packed || !y) return fail(GNERF_E_ARG, "conv3x3_epilogue_nhwc: null pointer");
    if (n < 1 || h < 1 || w < 1 || cin < 1 || cout < 1) return fail(GNERF_E_ARG, "conv3x3_epilogue_nhwc: empty tensor");
    if (h % kTH || w % kTW || cin % kCK || cout % kCO)
        return fail(GNERF_E_UNSUPPORTED, "conv3x3_epilogue_nhwc: needs height %% 8 == 0, width %% 32 == 0, input channels %% 64 == 0, output channels %% 128 == 0 (got %dx%d, %d -> %d)", h, w, cin, cout);
    if (size_t(h) * w * cin * 2 >= (size_t(1) << 31)) return fail(GNERF_E_UNSUPPORTED, "conv3x3_epilogue_nhwc: one image of x must stay below 2 GB");
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w_packed) | reinterpret_cast<uintptr_t>(y)) & 15)
        return fail(GNERF_E_ARG, "conv3x3_epilogue_nhwc: x, w and y must be 16-byte aligned");
    if ((scale && (reinterpret_cast<uintptr_t>(scale) & 15)) || (next_scale && (reinterpret_cast<uintptr_t>(next_scale) & 15)))
        return fail(GNERF_E_ARG, "conv3x3_epilogue_nhwc: scale and next_scale must be 16-byte aligned");
    ConvArgs a;
    a.x = static_cast<const _Float16*>(x); a.wpk = static_cast<const _Float16*>(w_packed); a.y = static_cast<_Float16*>(y);
    a.scale = scale; a.noise = noise; a.bias = static_cast<const __half*>(bias); a.next_scale = next_scale;
    a.n = n; a.h = h; a.w = w; a.cin = cin; a.cout = cout;
    a.tiles_x = w / kTW; a.tiles_y = h / kTH; a.groups = cout / kCO; a.n_jobs = n * a.tiles_x * a.tiles_y * a.groups;
    a.round_noise = round_noise; a.alpha = alpha; a.gain = gain; a.clamp = clamp;
    // persistent workgroups: one per CU (its LDS), fewer when there are fewer jobs; a multiple of the XCD count
    int wgs = a.n_jobs < kNumCU ? (a.n_jobs + kNumXCD - 1) / kNumXCD * kNumXCD : kNumCU;
    const dim3 grid(wgs), block(kConvThreads);
    hipStream_t s = as_stream(stream);
#define GNERF_CONV(SC, NZ, NX) do { \
        static bool raised[64] = {}; \
        int dev = 0; \
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0; \
        if (!raised[dev]) { \
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_epilogue_kernel<SC, NZ, NX>), hipFuncAttributeMaxDynamicSharedMemorySize, kConvLds) != hipSuccess) \
                return fail(GNERF_E_LAUNCH, "conv3x3_epilogue_nhwc: cannot raise the dynamic LDS limit"); \
            raised[dev] = true; \
        } \
        hipLaunchKernelGGL((conv3x3_epilogue_kernel<SC, NZ, NX>), grid, block, kConvLds, s, a); } while (0)
    const int key = (scale ? 4 : 0) | (noise ? 2 : 0) | (next_scale ? 1 : 0);
    switch (key) {
        case 0: GNERF_CONV(false, false, false); break;
        case 1: GNERF_CONV(false, false, true); break;
        case 2: GNERF_CONV(false, true, false); break;
        case 3: GNERF_CONV(false, true, true); break;
        case 4: GNERF_CONV(true, false, false); break;
        case 5: GNERF_CONV(true, false, true); break;
        case 6: GNERF_CONV(true, true, false); break;
        default: GNERF_CONV(true, true, true); break;
    }
#undef GNERF_CONV
    return check_launch("conv3x3_epilogue_nhwc");
}
