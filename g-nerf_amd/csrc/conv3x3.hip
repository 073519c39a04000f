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
// Decomposition.  A workgroup (round 6: 8 waves, <= 128 registers per lane, 76.5 KB of LDS, so TWO workgroups share a CU: FOUR waves per
// SIMD, and one workgroup's barriers, input-tile loads and epilogue hide under the other's MFMAs; rounds 5: 4 waves with 256 registers,
// GNERF_CONV_WAVES=4) computes an 8 x 32 tile of output pixels for 128 output channels.  The 10 x 34 input pixels the tile's nine taps touch are staged once per 64 input channels (43.5 KB; image borders
// come in as zeros from the buffer load's range check) and every tap reads them at a shifted position; the weights of one (tap, 64
// input channels) -- 16 KB -- stream through a double buffer while the previous tap is being multiplied.  Both are filled by LDS-DMA
// (buffer_load / global_load ... lds: no registers, no ds_write), whose LDS image is lane-linear, so the bank swizzle is applied to
// the SOURCE address and again when the fragments are read: in 128-byte rows, 16-byte slot ^ (pixel & 7) for the input tile and
// ^ ((channel >> 1) & 7) for the weights.  A ds_read_b128 is served in four groups of sixteen NON-contiguous lanes ({0-3, 12-15, 20-27},
// {4-11, 16-19, 28-31}, ... -- MI355X_MICROARCH.md, LDS): with lane = 16 * (k quarter) + row, a group is eight rows at one k slot and
// the eight rows between them at the next; tools/lds_bank_model.py evaluates the formulas against those groups (both conflict-free
// at every tile position; the weights' formula on the input tile is two-way at three positions of four, and was what round 5's
// first two-workgroup build ran: SQ_LDS_BANK_CONFLICT 640 cycles per wave of ~2 500).
// (History, profiles/r05_conv3x3_*.txt: one workgroup per CU with 128-channel chunks 0.49 ms; + software-pipelined fragment reads 0.42;
// persistent workgroups with a four-deep weight ring 0.62; this form 0.32 -- at one wave per SIMD every stall of the only wave is the
// SIMD's, and no hand-built pipeline beat two independent workgroups.)
// Orientation: A = weights (M = 16 output channels), B = input (N = 16 pixels of a row), K = 32 input channels per instruction; a lane's
// four accumulator registers are then four CONSECUTIVE output channels of one pixel -- an 8-byte piece of the channels_last result.
// A wave owns two rows of the tile and 64 of the 128 channels: 64 accumulator registers, 16 MFMAs per 8 ds_read_b128 (GNERF_CONV_WAVES=4: two
// rows x 128 channels, 128 accumulator registers, 32 MFMAs per 12 ds_read_b128, fragment reads software-pipelined among the MFMAs).

#include "common.h"
#include <type_traits>


namespace {

using namespace gnerf;

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int kTH = 8, kTW = 32;                    // output pixels of a tile
constexpr int kIH = kTH + 2, kIW = kTW + 2;         // input pixels incl. the one-pixel halo
constexpr int kCK = 64;                             // input channels resident in LDS (a chunk)
constexpr int kCO = 128;                            // output channels per workgroup
// GNERF_CONV_WAVES: waves of a workgroup.  4 (round 5): a wave owns two rows of the tile -- 128 accumulator registers, fragments double-buffered
// and read among the previous k-step's MFMAs, 252 registers: two waves per SIMD.  8 (round 6, shipped): 64 accumulator registers per wave,
// fragments single-buffered, <= 128 registers: FOUR waves per SIMD on the same 76 KB of LDS per workgroup, the latency hiding left to the
// hardware's wave scheduler instead of the hand-built pipeline.  In-kernel clock stamps (tools/conv_clock.py, profiles/r06_conv_clock.jsonl):
// 7-9 % fewer cycles than the four-wave form, of which the chip's power management takes most back as a lower clock (1.98 -> 1.85 GHz on
// random operands); wall time -1.5 % (plain) ... -9 % (transposed, fp32-grade at small sizes).
#ifndef GNERF_CONV_WAVES
#define GNERF_CONV_WAVES 8
#endif
// GNERF_CONV_COSPLIT (with eight waves): 1 = a wave owns one tile row x 128 output channels (8 weight + 2 pixel fragments per 16 MFMAs), 2
// (shipped) = two rows x 64 output channels (4 + 4 fragments per 16 MFMAs: a fifth fewer LDS bytes for the same matrix work; 1-2 % faster).
// GNERF_CONV_COUNTED_WAITS=1 (a pair of channel blocks starts when its own fragments have landed): measured, no gain, off.
#ifndef GNERF_CONV_COSPLIT
#define GNERF_CONV_COSPLIT (GNERF_CONV_WAVES == 8 ? 2 : 1)
#endif
#ifndef GNERF_CONV_COUNTED_WAITS
#define GNERF_CONV_COUNTED_WAITS 0
#endif
// GNERF_CONV_NT: 1 = the input tile's LDS-DMA loads non-temporal, 2 = the output stores non-temporal, 3 = both, 4 (shipped, round 6) = the plain
// convolution's output stores only.  The output is streamed (written once, read by the NEXT kernel) and should not evict the weights that every
// workgroup of an XCD re-reads from its L2: alone the kernel is 1-5 % faster with 2 (profiles/r06_conv_nt_ab.txt), in the orbit's pipeline -- the
// next kernel reads what was stored -- 8 views per call go from 2 838-2 856 to 2 879-2 884 frames/s.  The input tile is NOT stream-once (halos,
// the transposed form's four phases): 1 is 10-20 % slower on the transposed form; the transposed form's stores are a wash.
#ifndef GNERF_CONV_NT
#define GNERF_CONV_NT 4
#endif
constexpr int kWaves = GNERF_CONV_WAVES;
constexpr int kCoSplit = GNERF_CONV_COSPLIT;
static_assert(kWaves == 4 || kWaves == 8, "4 or 8 waves per workgroup");
static_assert(kCoSplit == 1 || (kCoSplit == 2 && kWaves == 8), "the output channels are split over wave pairs in the eight-wave form only");
constexpr int kConvThreads = 64 * kWaves;
constexpr int kRowsPW = kTH / (kWaves / kCoSplit);  // tile rows of a wave
constexpr int kPB = 2 * kRowsPW;                    // 16-pixel blocks of a wave
constexpr int kCB = 8 / kCoSplit;                   // 16-channel blocks of a wave
constexpr int kRow = kCK * 2;                       // bytes of a pixel's / an output channel's row in LDS: 8 slots of 16 bytes
constexpr int kXPieces = kIH * kIW * (kCK / 8);     // 16-byte pieces of the input tile
constexpr int kXRounds = (kXPieces + kConvThreads - 1) / kConvThreads;
constexpr int kXPiecesPad = (kXPieces + 63) / 64 * 64;      // whole waves of LDS-DMA (a wave writes 64 consecutive pieces); waves beyond them skip the last round
constexpr int kXBytes = kXPiecesPad * 16;
constexpr int kWBytes = kCO * kCK * 2;
constexpr int kWRounds = kWBytes / 16 / kConvThreads;
constexpr int kStorePix = kConvThreads / 16;        // pixels a trip of the store loop moves (16 slots of 16 bytes each)
constexpr int kOperands = kXBytes + 2 * kWBytes;                  // LDS offset of the epilogue's per-channel operands
constexpr int kConvLds = kOperands + 1536 + 768;                  // [128] float scale, [128] float next_scale, [128] half / float bias, RGB: [3][128] half ToRGB weights
static_assert(kTH * kTW * kCO * 2 <= kOperands, "the output tile is staged where the input tile and the weights were");
static_assert(2 * kConvLds <= 160 * 1024, "two workgroups per CU");
static_assert(kCK == 64, "two k-steps per tap; the swizzles assume 128-byte rows");

struct ConvArgs {
    const _Float16* x;          // [n, h, w, cin]    channels_last activations
    const _Float16* wpk;        // [9, cout, cin]     tap-major weights: tap = ky * 3 + kx of the correlation form
    _Float16* y;                // [n, h, w, cout]
    const float* scale;         // [n, cout] or NULL  (demodulation coefficients)
    const float* noise;         // [h * w] or NULL
    const __half* bias;         // [cout] or NULL
    const float* next_scale;    // [n, cout] or NULL
    float* y32;                 // OUT32: [n, h, w, cout] float32 result (y unused)
    const float* bias32;        // OUT32: [cout] float32 or NULL (bias unused)
    int n, h, w, cin, cout;
    int cin_pad;                // input channels of the packed weights: cin rounded up to a multiple of 64 (the extra channels are zeros)
    int tiles_x, tiles_y, n_tiles;
    int out_h, out_w;           // transposed form: 2 h + 1, 2 w + 1 (the convolution: h, w)
    int round_noise;
    int phase_jobs;             // transposed form: 0 = a workgroup computes all four phases of its position tile, 1 = ONE phase (blockIdx.z), 2 = two: {0, 3} or {1, 2}
    float alpha, gain, clamp;
    // RGB (ABI 11): the layer's ToRGB (networks_stylegan2.py:348-366, a modulated 1 x 1 convolution to three channels) taken in the epilogue, its result
    // ADDED to the block's running image -- no y is written at all (the superresolution's last block: nothing else reads its x)
    const _Float16* rgb_w;      // [n, 3, 128] float16: the ToRGB weight x its styles per sample (what gnerf_torgb_nhwc multiplies by)
    const float* rgb_bias;      // [3] float32 or NULL
    float rgb_clamp;            // < 0: none
    float* img;                 // [n, 3, h, w] float32, accumulated in place
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef unsigned u4nt __attribute__((ext_vector_type(4)));     // (what __builtin_nontemporal_store takes: GNERF_CONV_NT)
#ifndef GNERF_CONV_EPILOGUE_F32
#define GNERF_CONV_EPILOGUE_F32 1
#endif

// MODE 0: the 3x3 convolution (padding 1) with the epilogue.
// MODE 1 (round 5): the stride-2 TRANSPOSED 3x3 convolution of the x2 layers (conv_transpose2d(x, w, stride 2): 2H + 1 outputs per axis; what
// conv2d_resample.py:109-119 hands to the framework's transposed convolution), as its four output phases: output (2m + py, 2n + px) =
// sum over a in A(py), b in A(px) of x[m - a, n - b] . w[ky = py + 2a, kx = px + 2b], A(0) = {0, 1}, A(1) = {0} -- an ordinary
// convolution over the INPUT grid with 4, 2, 2 or 1 of the taps (dy, dx) = (1 - a, 1 - b) of the 3x3 stencil, so the staging, the
// fragment addressing and the MFMA loop are the convolution's; a job is (tile of 8 x 32 positions (m, n), phase), the result goes out
// as plain fp16 at stride 2 (the blur + epilogue pass that follows reads it).  Weights arrive packed per phase: [9][cout][cin] in the order
// phase (0,0): taps (a,b) = (0,0), (0,1), (1,0), (1,1); phase (0,1): a = 0, 1; phase (1,0): b = 0, 1; phase (1,1): the one tap.
// OUT32 (round 6): the fp32-GRADE form for the backbone's float32 layers (networks_stylegan2.py:41-98 on float32 activations).  The
// arithmetic is the fused renderer's decoder's: every product x w as hi(x) hi(w) + lo(x) hi(w) + hi(x) lo(w) with hi = f16(v), lo = f16(v - hi)
// and fp32 accumulation -- which for a convolution is this very kernel on three times the input channels, x' = [hi | lo | hi] (written by
// gnerf_split_f16x3_nhwc) against w' = [hi | hi | lo] (packed once per weight version).  What changes here is the way out: the epilogue's
// fp32 values are stored as they are (16 bytes per lane straight from the accumulators: a 256 x 128 fp32 tile does not fit the staging
// LDS), and the bias arrives as float32.  MIOpen's fp32 kernels reach 0.82-0.86 of the 157 TFLOP/s fp32 matrix peak on these shapes; three
// f16 matrix instructions per product at 16x the rate are 2.5-3.3x faster (profiles/r06_conv_f32grade_gate.jsonl).
#ifdef GNERF_CONV_STAMPS
// Diagnostic build (tools/build_variants.sh 'D:GNERF_CONV_STAMPS', tools/conv_clock.py): the first lane of every workgroup stamps the shader
// clock (s_memtime) and the constant 100 MHz clock (s_memrealtime) at its start and end; their quotient is the clock the chip HOLDS under this
// kernel (MI355X_MICROARCH.md, DVFS give-back item 6).  The stamps leave through a buffer of their own that nothing else reads.
__device__ unsigned long long g_conv_stamps[16384][8];
#define GNERF_CONV_STAMP(k) do { __builtin_amdgcn_sched_barrier(0); stamp_c[k] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define GNERF_CONV_STAMP(k) do { } while (0)
#endif

template <int MODE, bool SCALE, bool NOISE, bool NEXT, bool OUT32 = false, bool RGB = false>
__global__ __launch_bounds__(kConvThreads, kWaves == 8 ? 4 : 2) void conv3x3_epilogue_kernel(ConvArgs a) {
    static_assert(!RGB || (MODE == 0 && !OUT32 && !NEXT && GNERF_CONV_EPILOGUE_F32), "the ToRGB tail belongs to the plain fp16 convolution of a block's last layer");
    extern __shared__ __align__(16) char lds[];
#ifdef GNERF_CONV_STAMPS
    const unsigned long long stamp_c0 = __builtin_amdgcn_s_memtime(), stamp_r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long stamp_c[4] = {0, 0, 0, 0};                       // before the first loads / behind the prologue's barrier / at the main loop's end / in front of the store loop
#endif
    char* const xs = lds;                           // [340 pixels][8 slots of 16 bytes], slot ^= pixel & 7
    char* const wb = lds + kXBytes;                 // 2 x [128 output channels][8 slots], slot ^= (channel >> 1) & 7
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, hq = lane >> 4;
    const int wrow = wv / kCoSplit, wco = wv % kCoSplit;                // the wave's rows are kRowsPW * wrow ..., its channel blocks kCB * wco ...

    // tile of this workgroup: XCD b % 8 gets a contiguous eighth of the tile sequence (neighbouring tiles share halo pixels in its L2)
    const int per_xcd = (a.n_tiles + kNumXCD - 1) / kNumXCD;
    const int tile = (blockIdx.x % kNumXCD) * per_xcd + blockIdx.x / kNumXCD;
    if (tile >= a.n_tiles) return;
    const int tiles_img = a.tiles_x * a.tiles_y;
    const int H = a.h, W = a.w, Cin = a.cin, Cout = a.cout;
    // transposed form: a job is a position tile, and the workgroup walks its four phases one after the other (16, 8, 8, 4 steps): they read
    // the same input tile (L2 hits), every job is the same length, and three of four workgroup launches are gone.  (History,
    // profiles/r05_conv_transpose_phases.txt: a job per (phase, tile) with the phase slowest put ALL the four-tap jobs, 4/9 of the work, on
    // XCDs 0 and 1 -- an XCD owns a contiguous eighth of the job sequence -- : 0.287 ms; phase fastest: 0.262; a 4-step job costs 13 us on
    // its own, of which the MFMAs are 4.)
    const int tile_p = tile;
    // (round 6: the four phases are a ROLLED loop -- the taps of a phase, their shifts and the step counters below are scalar run-time
    //  values.  Unrolled, every phase had its own copy of the main loop and the epilogue, and the values the compiler hoisted across
    //  them cost the transposed form 34 spilled registers and 92 bytes of scratch per lane.)
    // phase_jobs (round 6): launches that do not fill the chip's 512 workgroup slots with whole position tiles -- the backbone's x2 layers from
    // 32^2 and 64^2 are 160 and 216 jobs -- run one phase per workgroup instead (blockIdx.z, the slowest grid index: the four-tap phase's jobs are
    // dispatched first and round-robin over the XCDs): four times the workgroups for the same work, jobs of 4 : 2 : 2 : 1 length
    // phase_jobs = 2 (round 6): phase PAIRS per workgroup -- {four taps, one tap} for blockIdx.z = 0 (dispatched first), {two taps, two taps} for 1:
    // jobs of 5 : 4 length, half as long as a whole position tile's.  A launch of J whole-tile jobs takes ceil(J / 512) rounds of the chip's 512
    // workgroup slots (4 x 256^2 positions: 1 188 jobs = 2.3 -> 3 rounds, the last one a third full); in halves it is 4.6 -> 5 rounds of half the length.
    const int n_ph = MODE == 1 ? (a.phase_jobs == 0 ? 4 : a.phase_jobs) : 1;
#pragma nounroll
    for (int pi = 0; pi < n_ph; pi++) {
    const int ph = MODE == 1 ? (a.phase_jobs == 0 ? pi : a.phase_jobs == 1 ? int(blockIdx.z) : (blockIdx.z == 0 ? 3 * pi : 1 + pi)) : 0;
    const int ph_y = ph >> 1, ph_x = ph & 1;                            // (py, px)
    const int n_taps = MODE == 1 ? (2 - ph_y) * (2 - ph_x) : 9;
    const int tap_base = MODE == 1 ? (ph == 0 ? 0 : ph == 1 ? 4 : ph == 2 ? 6 : 8) : 0;
    const int n = tile_p / tiles_img, tt = tile_p - n * tiles_img;
    const int ty = tt / a.tiles_x, tx = tt - ty * a.tiles_x;
    const int y0 = ty * kTH, x0 = tx * kTW;
    const int co0 = blockIdx.y * kCO;
    const int Hp = MODE == 1 ? H + 1 - ph_y : H, Wp = MODE == 1 ? W + 1 - ph_x : W;      // positions of this phase per axis
    if (MODE == 1 && (y0 >= Hp || x0 >= Wp)) continue;                 // a tile of the 8 x 32 grid over (H + 1) x (W + 1) that an odd phase does not have (uniform: no barrier is skipped by a part of the workgroup)
    // (dy, dx) of tap t of this job in the staged input tile (whose origin is one pixel up and left of the tile's first position)
    auto tap_shift = [&](int t, int& dy, int& dx) {
        if (MODE == 1) {
            const int ta = ph_y ? 0 : (ph_x ? t : t >> 1), tb = ph_x ? 0 : (ph_y ? t : t & 1);
            dy = 1 - ta; dx = 1 - tb;
        } else {
            dy = t / 3; dx = t - dy * 3;
        }
    };

    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<_Float16*>(a.x) + size_t(n) * H * W * Cin, 0, int(size_t(H) * W * Cin * 2), 0x00020000);

    auto stage_x = [&](int cin0) {
        // (the eleven source offsets depend on the lane and the tile only: left visible, the compiler computes them once in front of the
        //  main loop and keeps them -- or rather spills them, and a scratch reload in the loop waits for every LDS-DMA in flight)
        int t = tid;
        asm volatile("" : "+v"(t));
#pragma unroll
        for (int it = 0; it < kXRounds; it++) {
            if (it * kConvThreads + wv * 64 >= kXPiecesPad) break;      // (wave-uniform: the last round's waves past the tile)
            const int q = it * kConvThreads + t;
            const int pix = q >> 3, slot = q & 7;
            const int py = pix / kIW, px = pix - py * kIW;
            const int iy = y0 - 1 + py, ix = x0 - 1 + px;
            const int cg = cin0 + ((slot ^ (pix & 7)) << 3);        // first of this piece's eight input channels
            const bool ok = pix < kIH * kIW && iy >= 0 && iy < H && ix >= 0 && ix < W && cg < Cin;      // (cin below a multiple of 64: the last chunk's tail reads as zeros)
            const unsigned off = ok ? unsigned(((iy * W + ix) * Cin + cg) * 2) : 0x80000000u;     // out of range: zeros
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (lds_ptr_t)(xs + (it * kConvThreads + wv * 64) * 16), 16, off, 0, 0, (GNERF_CONV_NT & 1) ? 2 : 0);
        }
    };
    // weights of (tap, 64 input channels) -> buffer `buf`.  Lane (co = (threads / 8) it + tid / 8, slot = tid % 8) of trip `it` reads 16 bytes of output
    // channel co; the swizzle key (co >> 1) & 7 = (tid >> 4) & 7 does not depend on the trip, so the lane contributes ONE 32-bit offset and
    // everything else -- tap, chunk, trip -- is a scalar base (round 6: as four 64-bit lane pointers these were eight v_lshl_add_u64 per step)
    const unsigned w_lane = unsigned(((tid >> 3) * a.cin_pad + (((tid & 7) ^ ((tid >> 4) & 7)) << 3)) * 2);
    auto stage_w = [&](int tap, int cin0, int buf) {
        const char* src = reinterpret_cast<const char*>(a.wpk + (size_t(tap_base + tap) * Cout + co0) * a.cin_pad + cin0);
#pragma unroll
        for (int it = 0; it < kWRounds; it++) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + size_t(it) * (kConvThreads / 8) * a.cin_pad * 2 + w_lane),
                                             (lds_ptr_t)(wb + buf * kWBytes + (it * kConvThreads + wv * 64) * 16), 16, 0, 0);
        }
    };

    v4f acc[kCB][kPB];
#pragma unroll
    for (int cb = 0; cb < kCB; cb++)
#pragma unroll
        for (int pb = 0; pb < kPB; pb++) acc[cb][pb] = (v4f){0.f, 0.f, 0.f, 0.f};

    // The epilogue's per-channel operands (demodulation scale, next layer's scale, bias of this workgroup's 128 channels: 1.25 KB) come in
    // by LDS-DMA right here and are read from LDS at the end.  (As ordinary loads at the point of use they were up to 24 dependent
    // round trips at the end of every tile; ordinary loads issued early make hipcc put `s_waitcnt vmcnt(0)` between the LDS-DMA
    // instructions of the main loop -- its wait insertion does not count the two kinds of load apart.)
    char* const ep = lds + kOperands;
    if (wv < 2) {                                                  // (wave-uniform: threads 0..127, one float each)
        if constexpr (SCALE) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.scale + size_t(n) * Cout + co0 + tid),
                                                              (lds_ptr_t)(ep + wv * 256), 4, 0, 0);
        if constexpr (NEXT) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.next_scale + size_t(n) * Cout + co0 + tid),
                                                             (lds_ptr_t)(ep + 512 + wv * 256), 4, 0, 0);
    }
    if constexpr (OUT32) {
        if (wv < 2 && a.bias32) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.bias32 + co0 + tid),
                                                                 (lds_ptr_t)(ep + 1024 + wv * 256), 4, 0, 0);
    } else {
        if (wv == 0 && a.bias) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.bias + co0 + 2 * tid),
                                                                (lds_ptr_t)(ep + 1024), 4, 0, 0);
    }
    if constexpr (RGB) {                                           // [3][128] halves = 3 x 256 bytes, four bytes per lane
        if (wv < 3) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(reinterpret_cast<const char*>(a.rgb_w) + size_t(n) * 768 + wv * 256 + lane * 4),
                                                     (lds_ptr_t)(ep + 1536 + wv * 256), 4, 0, 0);
    }
    const int n_chunks = a.cin_pad / kCK, total = n_chunks * n_taps;
    // ---- main loop, software-pipelined ACROSS steps.  A step (one tap of one 64-channel chunk) is two k-steps of 32 MFMAs; while
    // the MFMAs of one k-step issue, the twelve fragments of the NEXT k-step are read -- the second k-step of this step, or the first
    // of the next step.  The one barrier of a step sits between its two k-steps: there the next step's weights (requested a step
    // ago) have landed, every wave has read the last of this step's weights (their buffer takes the step after next) and, at a
    // chunk's last tap, the last of the input tile (the next chunk is requested right there and arrives under 32 MFMAs).
    // (Round 5's first two-workgroup loop read a step's first twelve fragments at the step's top, behind the barrier: 2 % slower -- the
    // other workgroup of the CU already covered most of that; commit b13915a has both loops and the timing-only ablation macros.)
    h8 A[kWaves == 4 ? 2 : 1][kCB], B[kWaves == 4 ? 2 : 1][kPB];
    // fragment f of k-step kc of step s: f = 0..3 the input fragments of the wave's four pixel blocks, f = 4..11 the eight weight fragments.
    // Addresses from two lane constants and wave-uniform terms (the tap's shift, the weight buffer); pixel block f ^ 1 and the weight
    // fragments sit at constant offsets (16 pixels / 16 channels further: the swizzle's period is 8 rows), k-step 1 flips bit 6.
    const int xlane = (kRowsPW * wrow * kIW + r) * kRow;             // byte offset of this lane's pixel row (the wave's first tile row, column r) without the tap's shift
    const int alane = (wco * kCB * 16 + r) * kRow + ((hq ^ ((r >> 1) & 7)) << 4);
    // (tap, buf): the step's tap within its chunk and the parity of its weight buffer -- running counters of the loop below, no divisions
    auto load_frag = [&](int tap, int buf, int kc, int f, h8 (&Af)[kCB], h8 (&Bf)[kPB]) {
        if (f < kPB) {
            int dy, dx;
            tap_shift(tap, dy, dx);
            const int row = xlane + (((f >> 1) + dy) * kIW + dx) * kRow;          // pixel index * 128
            const int slot = (hq ^ (row >> 7)) & 7;
            Bf[f] = *reinterpret_cast<const h8*>(xs + ((row + (slot << 4)) ^ (kc << 6)) + (f & 1) * 16 * kRow);
        } else {
            Af[f - kPB] = *reinterpret_cast<const h8*>(wb + buf * kWBytes + (alane ^ (kc << 6)) + (f - kPB) * 16 * kRow);
        }
    };
    auto mfma_group = [&](int g, const h8 (&Af)[kCB], const h8 (&Bf)[kPB]) {
#pragma unroll
        for (int cb = 2 * g; cb < 2 * g + 2; cb++)
#pragma unroll
            for (int pb = 0; pb < kPB; pb++) acc[cb][pb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Af[cb], Bf[pb], acc[cb][pb], 0, 0, 0);
    };
    // 32 MFMAs from (Ac, Bc) with the twelve reads of (s2, kc2) into (An, Bn) among them: {8 MFMAs, six reads} twice, then 16 MFMAs --
    // fenced so that the compiler keeps the order; the last read has 24 MFMAs to come back behind
    auto phase = [&](const h8 (&Ac)[kCB], const h8 (&Bc)[kPB], bool reads, int tap2, int buf2, int kc2, h8 (&An)[kCB], h8 (&Bn)[kPB]) {
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(0, Ac, Bc);
        __builtin_amdgcn_sched_barrier(0);
        if (reads) {
#pragma unroll
            for (int f = 0; f < 6; f++) load_frag(tap2, buf2, kc2, f, An, Bn);
        }
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(1, Ac, Bc);
        __builtin_amdgcn_sched_barrier(0);
        if (reads) {
#pragma unroll
            for (int f = 6; f < 12; f++) load_frag(tap2, buf2, kc2, f, An, Bn);
        }
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(2, Ac, Bc);
        mfma_group(3, Ac, Bc);
        __builtin_amdgcn_sched_barrier(0);
    };
    // step s = (chunk, tap); s + 1 and s + 2 are kept alongside as running counters
    auto next_step = [&](int& tp, int& ck) { if (++tp == n_taps) { tp = 0; ck++; } };
    int tap = 0, chunk = 0, tap1 = 0, chunk1 = 0, tap2, chunk2;
    next_step(tap1, chunk1);
    tap2 = tap1; chunk2 = chunk1;
    next_step(tap2, chunk2);
    GNERF_CONV_STAMP(0);
    stage_x(0);
    stage_w(0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    GNERF_CONV_STAMP(1);
    if (total > 1) stage_w(tap1, chunk1 * kCK, 1);
    if constexpr (kWaves == 4) {
#pragma unroll
    for (int f = 0; f < 12; f++) load_frag(0, 0, 0, f, A[0], B[0]);
    for (int s = 0; s < total; s++) {
        phase(A[0], B[0], true, tap, s & 1, 1, A[1], B[1]);
        // (bare waits and barrier: __syncthreads() would drain vmcnt where the compiler sees fit)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const bool more = s + 1 < total, new_chunk = more && tap == n_taps - 1;
        if (s + 2 < total) stage_w(tap2, chunk2 * kCK, s & 1);
        if (new_chunk) stage_x((chunk + 1) * kCK);
        phase(A[1], B[1], more && !new_chunk, tap1, (s + 1) & 1, 0, A[0], B[0]);
        if (new_chunk) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
#pragma unroll
            for (int f = 0; f < 12; f++) load_frag(tap1, (s + 1) & 1, 0, f, A[0], B[0]);
        }
        tap = tap1; chunk = chunk1; tap1 = tap2; chunk1 = chunk2;
        next_step(tap2, chunk2);
    }
    } else {
    // Eight waves: per k-step the wave reads its ten fragments, waits for them and issues its sixteen MFMAs; nothing is double-buffered in
    // registers -- three other waves of the SIMD have MFMAs to issue while this one waits.  The step's barrier sits where BOTH k-steps'
    // fragments of this step's weight buffer are in registers (behind the second read): the buffer then takes the step after next.
    for (int s = 0; s < total; s++) {
#pragma unroll
        for (int f = 0; f < kCB + kPB; f++) load_frag(tap, s & 1, 0, f, A[0], B[0]);
#if GNERF_CONV_COUNTED_WAITS
        // (the reads return in order: a pair of channel blocks starts as soon as ITS two weight fragments are there)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < kCB / 2; g++) {
            if (kCB - 2 - 2 * g == 6) asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
            else if (kCB - 2 - 2 * g == 4) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
            else if (kCB - 2 - 2 * g == 2) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            mfma_group(g, A[0], B[0]);
            __builtin_amdgcn_sched_barrier(0);
        }
#else
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < kCB / 2; g++) mfma_group(g, A[0], B[0]);
        __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
        for (int f = 0; f < kCB + kPB; f++) load_frag(tap, s & 1, 1, f, A[0], B[0]);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const bool more = s + 1 < total, new_chunk = more && tap == n_taps - 1;
        if (s + 2 < total) stage_w(tap2, chunk2 * kCK, s & 1);
        if (new_chunk) stage_x((chunk + 1) * kCK);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < kCB / 2; g++) mfma_group(g, A[0], B[0]);
        __builtin_amdgcn_sched_barrier(0);
        if (new_chunk) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        tap = tap1; chunk = chunk1; tap1 = tap2; chunk1 = chunk2;
        next_step(tap2, chunk2);
    }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    GNERF_CONV_STAMP(2);
    __syncthreads();                                                // the input tile is dead: its LDS takes the output tile

    // ---- epilogue in registers, then through LDS for 16-byte coalesced stores.  Output image: [256 pixels][16 slots], slot ^= pixel & 15.
    char* const os = lds;
    // (two copies of the register part, chosen once per workgroup: with the clamp's presence a run-time value the compiler keeps a
    //  compare-and-select per element behind every v_med3_f32 -- 128 of a wave's ~2 000 epilogue instructions)
#ifdef GNERF_ABLATE_CONVNOSTORE
    unsigned ablate_sink = 0;
#endif
    auto registers_to_lds = [&](auto has_clamp) {
    const float clampv = decltype(has_clamp)::value ? fabsf(a.clamp) : -1.f;
    if constexpr (decltype(has_clamp)::value) __builtin_assume(clampv >= 0.f);
#if GNERF_CONV_EPILOGUE_F32
    // Round 6: the epilogue in fp32 on the accumulators, ONE rounding at the end.
    //     y = clamp(lrelu((acc * dcoef + noise + bias)) * gain) * next_scale
    //       = med3(max(u, alpha u), -c, c) * next_scale,   u = acc * (dcoef gain) + (bias gain) + (noise gain)       (gain > 0, 0 <= alpha <= 1)
    // Per value: one fused multiply-add, one add, the lrelu's multiply and max, the clamp's v_med3, the next layer's multiply -- the three
    // fp32 multiply / add steps as packed instructions on accumulator pairs -- and half a v_cvt_pk_f16_f32: 4.5 vector instructions where
    // the form that reproduced the two-launch route's fp16 roundings (conv output, demodulated value, activated value: GNERF_CONV_EPILOGUE_F32=0)
    // took 18, most of them conversions.  Against the reference's fp32 arithmetic (networks_stylegan2.py:41-98, bias_act.py:92-122) this is
    // CLOSER than the fp16 chain it replaces: the only rounding left is the output's.  Per-channel operands are scaled by the gain once per
    // channel block, the lane's four noise values once per tile.
    typedef float v2f __attribute__((ext_vector_type(2)));
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const float g = a.gain;
    int lr = r;
    asm volatile("" : "+v"(lr));                                       // (per phase, see the store loop)
    const unsigned os_lane = unsigned(lr * 256 + (((hq >> 1) ^ lr) << 4) + (hq & 1) * 8);
    const unsigned y32_lane = unsigned((MODE == 1 ? 2 * lr : lr) * Cout + hq * 4);
    float nzg[kPB] = {};
    if constexpr (NOISE && MODE == 0) {
#pragma unroll
        for (int pb = 0; pb < kPB; pb++) {
            float nz = a.noise[(y0 + kRowsPW * wrow + (pb >> 1)) * W + x0 + (pb & 1) * 16 + r];
            if (a.round_noise) nz = round_to<__half>(nz);             // (the caller's noise tensor was fp16: noise.to(x.dtype), networks_stylegan2.py:313)
            nzg[pb] = nz * g;
        }
    }
#pragma unroll
    for (int cbw = 0; cbw < kCB; cbw++) {
        const int cb = wco * kCB + cbw;                            // (the channel block among the workgroup's eight)
        const int c4 = cb * 16 + hq * 4;                           // this lane's four consecutive output channels (of the workgroup's 128)
        v2f scg[2] = {{g, g}, {g, g}}, nx[2] = {{1.f, 1.f}, {1.f, 1.f}}, bg[2] = {{0.f, 0.f}, {0.f, 0.f}};
        if constexpr (SCALE) { const float4 v = *reinterpret_cast<const float4*>(ep + c4 * 4); scg[0] = (v2f){v.x, v.y} * g; scg[1] = (v2f){v.z, v.w} * g; }
        if constexpr (NEXT) { const float4 v = *reinterpret_cast<const float4*>(ep + 512 + c4 * 4); nx[0] = (v2f){v.x, v.y}; nx[1] = (v2f){v.z, v.w}; }
        if constexpr (OUT32) {
            if (a.bias32) { const float4 v = *reinterpret_cast<const float4*>(ep + 1024 + c4 * 4); bg[0] = (v2f){v.x, v.y} * g; bg[1] = (v2f){v.z, v.w} * g; }
        } else if (a.bias) {
            typedef _Float16 h4v __attribute__((ext_vector_type(4)));
            const h4v hb = __builtin_bit_cast(h4v, *reinterpret_cast<const uint2*>(ep + 1024 + c4 * 2));
            bg[0] = (v2f){float(hb[0]), float(hb[1])} * g; bg[1] = (v2f){float(hb[2]), float(hb[3])} * g;
        }
#pragma unroll
        for (int pb = 0; pb < kPB; pb++) {
            const int prow = kRowsPW * wrow + (pb >> 1);              // (the pixel: row prow, column (pb & 1) * 16 + r of the tile)
            unsigned words[2];
            float full[4];
#pragma unroll
            for (int k2 = 0; k2 < 2; k2++) {
                const v2f av = {acc[cbw][pb][2 * k2], acc[cbw][pb][2 * k2 + 1]};
                h2 out;
                if constexpr (MODE == 1) {
                    out = (h2){(_Float16)av[0], (_Float16)av[1]};      // the transposed convolution leaves as plain fp16 (blur + epilogue follow)
                    full[2 * k2] = av[0]; full[2 * k2 + 1] = av[1];
                } else {
                    v2f u = __builtin_elementwise_fma(av, scg[k2], bg[k2]);
                    if constexpr (NOISE) u = u + (v2f){nzg[pb], nzg[pb]};
                    const v2f ua = u * a.alpha;
                    // lrelu with a slope in [0, 1] (checked by the launcher) = max(u, alpha u).  The bare instruction: fmaxf() on the packed
                    // operations' results costs a canonicalising v_max_f32 x, x apiece in IEEE mode (256 of them per lane here); the
                    // instruction itself returns the other operand for a NaN, like fmaxf
                    float r0, r1;
                    asm("v_max_f32 %0, %1, %2" : "=v"(r0) : "v"(u[0]), "v"(ua[0]));
                    asm("v_max_f32 %0, %1, %2" : "=v"(r1) : "v"(u[1]), "v"(ua[1]));
                    if constexpr (decltype(has_clamp)::value) { r0 = __builtin_amdgcn_fmed3f(r0, -clampv, clampv); r1 = __builtin_amdgcn_fmed3f(r1, -clampv, clampv); }
                    v2f rr = {r0, r1};
                    if constexpr (NEXT) rr = rr * nx[k2];
                    out = (h2){(_Float16)rr[0], (_Float16)rr[1]};
                    full[2 * k2] = rr[0]; full[2 * k2 + 1] = rr[1];
                }
                words[k2] = __builtin_bit_cast(unsigned, out);
            }
            if constexpr (OUT32) {
                // four consecutive channels of one pixel = 16 bytes, straight from the accumulators (the lanes of a 16-lane row group cover 64
                // contiguous bytes of a pixel, channel blocks cb and cb ^ 1 its 128-byte line)
                const int yy = y0 + prow, xc = x0 + (pb & 1) * 16;      // (uniform) the pixel's row, the first column of its block of sixteen
                const float4 o4 = make_float4(full[0], full[1], full[2], full[3]);
                if constexpr (MODE == 1) {
                    float* const row = a.y32 + ((size_t(n) * a.out_h + size_t(2 * yy + ph_y)) * a.out_w + size_t(2 * xc + ph_x)) * Cout + co0 + cb * 16;
                    if (yy < Hp && xc + lr < Wp) *reinterpret_cast<float4*>(row + y32_lane) = o4;
                } else {
                    float* const row = a.y32 + (size_t(n) * H * W + size_t(yy) * W + xc) * Cout + co0 + cb * 16;
                    *reinterpret_cast<float4*>(row + y32_lane) = o4;
                }
                continue;
            }
            // (p = 32 prow + 16 (pb & 1) + r: the lane's part of the address is r * 256 + its slot, the rest is wave-uniform; c4 >> 3 = 2 cb +
            //  (hq >> 1) and 2 cb only touches the slot's upper bits, so channel block cb is the cb = 0 address with cb << 5 XORed in)
#ifdef GNERF_ABLATE_CONVNOSTORE  // timing-only build: the epilogue's values go nowhere (no staging, no store loop): what a consumer fused into the epilogue could save
            ablate_sink ^= words[0] ^ words[1];
            (void)os_lane;
#else
            *reinterpret_cast<uint2*>(os + ((os_lane ^ unsigned(cb << 5)) + unsigned(prow * 8192 + (pb & 1) * 4096))) = make_uint2(words[0], words[1]);
#endif
        }
    }
#else
#pragma unroll
    for (int cbw = 0; cbw < kCB; cbw++) {
        const int cb = wco * kCB + cbw;
        const int c4 = cb * 16 + hq * 4;                           // this lane's four consecutive output channels (of the workgroup's 128)
        float sc[4] = {1.f, 1.f, 1.f, 1.f}, nx[4] = {1.f, 1.f, 1.f, 1.f}, bv[4] = {0.f, 0.f, 0.f, 0.f};
        if constexpr (SCALE) { const float4 v = *reinterpret_cast<const float4*>(ep + c4 * 4); sc[0] = v.x; sc[1] = v.y; sc[2] = v.z; sc[3] = v.w; }
        if constexpr (NEXT) {
            const float4 v = *reinterpret_cast<const float4*>(ep + 512 + c4 * 4);
            nx[0] = round_to<__half>(v.x); nx[1] = round_to<__half>(v.y); nx[2] = round_to<__half>(v.z); nx[3] = round_to<__half>(v.w);
        }
        if (a.bias) {
            typedef _Float16 h4v __attribute__((ext_vector_type(4)));
            const h4v hb = __builtin_bit_cast(h4v, *reinterpret_cast<const uint2*>(ep + 1024 + c4 * 2));
#pragma unroll
            for (int k = 0; k < 4; k++) bv[k] = float(hb[k]);
        }
#pragma unroll
        for (int pb = 0; pb < kPB; pb++) {
            const int prow = kRowsPW * wrow + (pb >> 1), pcol = (pb & 1) * 16 + r;
            const int p = prow * kTW + pcol;
            Pk<__half, 4> in;
#pragma unroll
            for (int k = 0; k < 4; k++) in.v[k] = __float2half(acc[cbw][pb][k]);          // what the convolution alone would have stored
            float nz = 0.f;
            if constexpr (NOISE) nz = a.noise[(y0 + prow) * W + x0 + pcol];
            Pk<__half, 4> out;
            if constexpr (MODE == 1) out = in;                         // the transposed convolution leaves as plain fp16 (blur + epilogue follow)
            else out = modconv_epilogue_vec<__half, 4, kActLrelu01, SCALE, NOISE, NEXT>(in, sc, nz, a.round_noise != 0, bv, nx, a.alpha, a.gain, clampv);
            *reinterpret_cast<uint2*>(os + p * 256 + (((c4 >> 3) ^ (p & 15)) << 4) + ((c4 >> 2) & 1) * 8) = __builtin_bit_cast(uint2, out);
        }
    }
#endif
    };
    if (MODE == 0 && a.clamp >= 0.f) registers_to_lds(std::true_type{}); else registers_to_lds(std::false_type{});
#ifdef GNERF_ABLATE_CONVNOSTORE
    if (MODE == 0 && !OUT32 && !RGB) { if (ablate_sink == 0x12345679u) a.y[tid] = (_Float16)0; continue; }
#endif
    if constexpr (RGB) {
        // ToRGB from the STAGED tile (the layer's fp16 result, as the stand-alone ToRGB reads it from memory -- gnerf_torgb_nhwc_accumulate,
        // csrc/modconv.hip: f16 weights x styles, v_dot2_f32_f16, the sum rounded to fp16, bias + clamp, rounded again, added to the fp32 image):
        // two lanes per pixel, 64 channels each = eight 16-byte slots of the pixel's swizzled 256-byte row; the weights, [3][128] halves, are
        // read from LDS as broadcasts.  A wave covers one tile row: a plane's 32 consecutive pixels are 128 contiguous bytes of the NCHW image.
        // (From the accumulators instead -- every value multiplied into three per-lane sums before it is rounded -- the epilogue, already at the
        // register limit, spilled 70-93 registers.)
        __syncthreads();
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        const char* const wrgb = ep + 1536;
        for (int t2 = tid; t2 < 2 * kTH * kTW; t2 += kConvThreads) {
            const int pp = t2 >> 1, half = t2 & 1;
            float acc3[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int sl = 0; sl < 8; sl++) {
                const int slot = 8 * half + sl;
                const uint4 xv = *reinterpret_cast<const uint4*>(os + pp * 256 + ((slot ^ (pp & 15)) << 4));
                const unsigned xq[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const uint4 wv4 = *reinterpret_cast<const uint4*>(wrgb + k * 256 + slot * 16);
                    const unsigned wq[4] = {wv4.x, wv4.y, wv4.z, wv4.w};
#pragma unroll
                    for (int e = 0; e < 4; e++) acc3[k] = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, xq[e]), __builtin_bit_cast(h2, wq[e]), acc3[k], false);
                }
            }
#pragma unroll
            for (int k = 0; k < 3; k++) acc3[k] += __shfl_xor(acc3[k], 1);
            if (half == 0) {
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    float rv = round_to<__half>(acc3[k]) + (a.rgb_bias ? a.rgb_bias[k] : 0.f);
                    if (a.rgb_clamp >= 0.f) rv = __builtin_amdgcn_fmed3f(rv, -a.rgb_clamp, a.rgb_clamp);
                    float* const dst = a.img + ((size_t(n) * 3 + k) * H + y0 + pp / kTW) * W + x0 + pp % kTW;
                    *dst += round_to<__half>(rv);
                }
            }
        }
        continue;
    }
    if constexpr (OUT32) {
        if (MODE == 1) __syncthreads();                              // (uniform) the next phase's input tile must not land while a wave still reads this one's operands
        continue;
    }
    __syncthreads();
    {
        // The lanes move kStorePix (16 or 32) pixels x 16 slots per trip: pixel it * kStorePix + lane / 16 of the tile, slot lane % 16.  Everything that
        // depends on the trip is wave-uniform (a scalar base pointer, an immediate LDS offset); the lane contributes ONE 32-bit offset to each
        // side.  (Written as 64-bit addresses per trip the sixteen of them were hoisted out of the phase loop and spilled.)
        GNERF_CONV_STAMP(3);
        int lp = tid >> 4;
        asm volatile("" : "+v"(lp));                                   // (per phase: nothing of this is kept across the main loop)
        const int slot = tid & 15;
        const unsigned lds_lane = unsigned(lp * 256 + ((slot ^ (lp & 15)) << 4));
        const unsigned out_lane = unsigned((MODE == 1 ? 2 * lp : lp) * Cout + slot * 8);
#pragma unroll
        for (int it = 0; it < kTH * kTW * 16 / kConvThreads; it++) {
            const uint4 v = *reinterpret_cast<const uint4*>(os + lds_lane + it * kStorePix * 256);
            const int yy = y0 + it * kStorePix / kTW, xc = x0 + it * kStorePix % kTW;     // (uniform) row of the tile, first of this trip's columns
            if constexpr (MODE == 1) {                                 // position (yy, xx) of phase (py, px) -> output pixel (2 yy + py, 2 xx + px); the grid of tiles overhangs
                _Float16* const row = a.y + ((size_t(n) * a.out_h + size_t(2 * yy + ph_y)) * a.out_w + size_t(2 * xc + ph_x)) * Cout + co0;
                if (yy < Hp && xc + lp < Wp) { if (GNERF_CONV_NT & 2) __builtin_nontemporal_store(__builtin_bit_cast(u4nt, v), reinterpret_cast<u4nt*>(row + out_lane)); else *reinterpret_cast<uint4*>(row + out_lane) = v; }
            } else {
                _Float16* const row = a.y + (size_t(n) * H * W + size_t(yy) * W + xc) * Cout + co0;
                if (GNERF_CONV_NT & 6) __builtin_nontemporal_store(__builtin_bit_cast(u4nt, v), reinterpret_cast<u4nt*>(row + out_lane)); else *reinterpret_cast<uint4*>(row + out_lane) = v;
            }
        }
    }
#ifdef GNERF_CONV_STAMPS
    if (threadIdx.x == 0 && pi == n_ph - 1) {
        const unsigned slot = (blockIdx.y * gridDim.x + blockIdx.x) & 16383u;
        g_conv_stamps[slot][0] = stamp_c0; g_conv_stamps[slot][1] = __builtin_amdgcn_s_memtime();
        g_conv_stamps[slot][2] = stamp_r0; g_conv_stamps[slot][3] = __builtin_amdgcn_s_memrealtime();
        for (int k = 0; k < 4; k++) g_conv_stamps[slot][4 + k] = stamp_c[k];
    }
#endif
    if (MODE == 1) __syncthreads();                                  // the next phase's input tile overwrites the staged output
    // (stores straight from the accumulators -- 8 bytes per lane, no LDS staging, no barrier pair -- were measured: 0.237 -> 0.247 ms, the
    //  partial lines cost more than the staging)
    }
}

// x [n, pixels, c] float32 channels_last, scale [n, c] or NULL -> y [n, pixels, 3 c] float16 = [hi | lo | hi] of v = x * scale, hi = f16(v),
// lo = f16(v - hi): the activation operand of the fp32-grade convolution (OUT32 above).  A lane owns eight channels of a pixel: 32 bytes
// in, three 16-byte pieces out; a workgroup's lanes walk the pixels of one image with a grid stride that is a multiple of the lanes per
// pixel, so a lane meets the same channels every trip and keeps its eight scales in registers.  |v| beyond f16's range saturates to the
// largest finite half (the products then are wrong but finite) and raises the sticky word `overflow` (one atomic per workgroup that saw
// one): the convolution's range is f16's, like the renderer's f16 decoder body, and callers that cannot bound their activations check it.
__global__ __launch_bounds__(256) void split_f16x3_kernel(const float* __restrict__ x, const float* __restrict__ scale, _Float16* __restrict__ y,
                                                           int pixels, int c, int* overflow) {
    const int n = blockIdx.y;
    const int per_pixel = c >> 3;                                   // lanes per pixel
    const int64_t total = int64_t(pixels) * per_pixel;
    const int64_t stride = int64_t(gridDim.x) * 256 / per_pixel * per_pixel;         // a multiple of per_pixel: the channel group of a lane never changes
    int64_t q = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (q >= stride) return;                                        // (the lanes beyond the last whole pixel group of the stride)
    const int cg = int(q % per_pixel) * 8;
    float sc[8];
#pragma unroll
    for (int k = 0; k < 8; k++) sc[k] = scale ? scale[size_t(n) * c + cg + k] : 1.f;
    const float* xn = x + size_t(n) * pixels * c;
    _Float16* yn = y + size_t(n) * pixels * 3 * c;
    bool over = false;
    typedef _Float16 h8v __attribute__((ext_vector_type(8)));
    for (; q < total; q += stride) {
        const int64_t pix = q / per_pixel;
        const float4 a0 = *reinterpret_cast<const float4*>(xn + pix * c + cg), a1 = *reinterpret_cast<const float4*>(xn + pix * c + cg + 4);
        const float v[8] = {a0.x * sc[0], a0.y * sc[1], a0.z * sc[2], a0.w * sc[3], a1.x * sc[4], a1.y * sc[5], a1.z * sc[6], a1.w * sc[7]};
        h8v hi, lo;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const float vc = __builtin_amdgcn_fmed3f(v[k], -65504.f, 65504.f);
            over = over || vc != v[k];                              // (a NaN compares unequal too: it is reported, and travels on as a NaN)
            const float vv = v[k] == v[k] ? vc : v[k];
            hi[k] = (_Float16)vv;
            lo[k] = (_Float16)(vv - float(hi[k]));
        }
        _Float16* o = yn + pix * 3 * c + cg;
        *reinterpret_cast<h8v*>(o) = hi;
        *reinterpret_cast<h8v*>(o + c) = lo;
        *reinterpret_cast<h8v*>(o + 2 * c) = hi;
    }
    if (overflow && __any(over) && (threadIdx.x & 63) == 0) atomicOr(overflow, 1);
}

}  // namespace

// y = epilogue(conv3x3(x, w)) on float16 channels_last tensors; see include/gnerf_hip.h.
namespace {
template <bool OUT32>
int launch_conv3x3(const char* what, const void* x, const void* w_packed, void* y, int n, int h, int w, int cin, int cout,
                   const float* scale, const float* noise, int round_noise, const void* bias,
                   float alpha, float gain, float clamp, const float* next_scale, gnerf_stream_t stream,
                   const void* rgb_w = nullptr, const float* rgb_bias = nullptr, float rgb_clamp = -1.f, float* img = nullptr) {
    using namespace gnerf;
    if (img) {
        if (OUT32 || !rgb_w || !scale || next_scale || cout != kCO)
            return fail(GNERF_E_UNSUPPORTED, "%s: the ToRGB tail needs 128 output channels, a demodulation scale and no next-layer scale", what);
        if ((reinterpret_cast<uintptr_t>(rgb_w) | reinterpret_cast<uintptr_t>(img)) & 3) return fail(GNERF_E_ARG, "%s: rgb_w and img must be 4-byte aligned", what);
        y = const_cast<void*>(x);                                   // (never written: the checks below want a pointer)
    }
    if (!x || !w_packed || !y) return fail(GNERF_E_ARG, "%s: null pointer", what);
    if (n < 1 || h < 1 || w < 1 || cin < 1 || cout < 1) return fail(GNERF_E_ARG, "%s: empty tensor", what);
    if (h % kTH || w % kTW || cin % 8 || cout % kCO)
        return fail(GNERF_E_UNSUPPORTED, "%s: needs height %% 8 == 0, width %% 32 == 0, input channels %% 8 == 0, output channels %% 128 == 0 (got %dx%d, %d -> %d)", what, h, w, cin, cout);
    if (size_t(h) * w * cin * 2 >= (size_t(1) << 31)) return fail(GNERF_E_UNSUPPORTED, "%s: one image of x must stay below 2 GB", what);
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w_packed) | reinterpret_cast<uintptr_t>(y)) & 15)
        return fail(GNERF_E_ARG, "%s: x, w and y must be 16-byte aligned", what);
    if ((scale && (reinterpret_cast<uintptr_t>(scale) & 15)) || (next_scale && (reinterpret_cast<uintptr_t>(next_scale) & 15)))
        return fail(GNERF_E_ARG, "%s: scale and next_scale must be 16-byte aligned", what);
    if (!(alpha >= 0.f && alpha <= 1.f)) return fail(GNERF_E_UNSUPPORTED, "%s: the lrelu slope must lie in [0, 1] (got %g)", what, double(alpha));
    if (!(gain > 0.f)) return fail(GNERF_E_UNSUPPORTED, "%s: the gain must be positive (got %g): it is folded into the lrelu's operands", what, double(gain));
    if (bias && (reinterpret_cast<uintptr_t>(bias) & 3)) return fail(GNERF_E_ARG, "%s: bias must be 4-byte aligned", what);
    ConvArgs a;
    a.x = static_cast<const _Float16*>(x); a.wpk = static_cast<const _Float16*>(w_packed);
    a.y = OUT32 ? nullptr : static_cast<_Float16*>(y); a.y32 = OUT32 ? static_cast<float*>(y) : nullptr;
    a.scale = scale; a.noise = noise; a.next_scale = next_scale;
    a.bias = OUT32 ? nullptr : static_cast<const __half*>(bias); a.bias32 = OUT32 ? static_cast<const float*>(bias) : nullptr;
    a.n = n; a.h = h; a.w = w; a.cin = cin; a.cout = cout;
    a.cin_pad = (cin + kCK - 1) / kCK * kCK;
    a.tiles_x = w / kTW; a.tiles_y = h / kTH; a.n_tiles = n * a.tiles_x * a.tiles_y;
    a.out_h = h; a.out_w = w;
    a.round_noise = round_noise; a.alpha = alpha; a.gain = gain; a.clamp = clamp; a.phase_jobs = 0;
    a.rgb_w = static_cast<const _Float16*>(rgb_w); a.rgb_bias = rgb_bias; a.rgb_clamp = rgb_clamp; a.img = img;
    const dim3 grid((a.n_tiles + kNumXCD - 1) / kNumXCD * kNumXCD, cout / kCO), block(kConvThreads);
    hipStream_t s = as_stream(stream);
#if !GNERF_CONV_EPILOGUE_F32
    if (img) return fail(GNERF_E_UNSUPPORTED, "%s: this build (GNERF_CONV_EPILOGUE_F32=0) has no ToRGB tail", what);
#else
    if constexpr (!OUT32) {
        if (img) {
            static PerDeviceOnce once_rgb[2];
            if (noise) {
                if (int rc = once_rgb[1].raise_lds(conv3x3_epilogue_kernel<0, true, true, false, false, true>, what, kConvLds)) return rc;
                hipLaunchKernelGGL((conv3x3_epilogue_kernel<0, true, true, false, false, true>), grid, block, kConvLds, s, a);
            } else {
                if (int rc = once_rgb[0].raise_lds(conv3x3_epilogue_kernel<0, true, false, false, false, true>, what, kConvLds)) return rc;
                hipLaunchKernelGGL((conv3x3_epilogue_kernel<0, true, false, false, false, true>), grid, block, kConvLds, s, a);
            }
            return check_launch(what);
        }
    }
#endif
#define GNERF_CONV(SC, NZ, NX) do { \
        static PerDeviceOnce once; \
        if (int rc = once.raise_lds(conv3x3_epilogue_kernel<0, SC, NZ, NX, OUT32>, what, kConvLds)) return rc; \
        hipLaunchKernelGGL((conv3x3_epilogue_kernel<0, SC, NZ, NX, OUT32>), grid, block, kConvLds, s, a); } while (0)
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
    return check_launch(what);
}

template <bool OUT32>
int launch_conv_transpose(const char* what, const void* x, const void* w_phases, void* y, int n, int h, int w, int cin, int cout, gnerf_stream_t stream) {
    using namespace gnerf;
    if (!x || !w_phases || !y) return fail(GNERF_E_ARG, "%s: null pointer", what);
    if (n < 1 || h < 1 || w < 1 || cin < 1 || cout < 1) return fail(GNERF_E_ARG, "%s: empty tensor", what);
    if (cin % 8 || cout % kCO)
        return fail(GNERF_E_UNSUPPORTED, "%s: needs input channels %% 8 == 0, output channels %% 128 == 0 (got %d -> %d)", what, cin, cout);
    if (size_t(h) * w * cin * 2 >= (size_t(1) << 31)) return fail(GNERF_E_UNSUPPORTED, "%s: one image of x must stay below 2 GB", what);
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w_phases) | reinterpret_cast<uintptr_t>(y)) & 15)
        return fail(GNERF_E_ARG, "%s: x, w and y must be 16-byte aligned", what);
    ConvArgs a;
    a.x = static_cast<const _Float16*>(x); a.wpk = static_cast<const _Float16*>(w_phases);
    a.y = OUT32 ? nullptr : static_cast<_Float16*>(y); a.y32 = OUT32 ? static_cast<float*>(y) : nullptr;
    a.scale = nullptr; a.noise = nullptr; a.bias = nullptr; a.bias32 = nullptr; a.next_scale = nullptr;
    a.rgb_w = nullptr; a.rgb_bias = nullptr; a.rgb_clamp = -1.f; a.img = nullptr;
    a.n = n; a.h = h; a.w = w; a.cin = cin; a.cout = cout;
    a.cin_pad = (cin + kCK - 1) / kCK * kCK;
    a.tiles_x = (w + 1 + kTW - 1) / kTW; a.tiles_y = (h + 1 + kTH - 1) / kTH;         // tiles over the (h + 1) x (w + 1) positions of the even phases
    const int64_t jobs = int64_t(n) * a.tiles_x * a.tiles_y;
    if (jobs > (int64_t(1) << 30)) return fail(GNERF_E_UNSUPPORTED, "%s: too many tiles", what);
    a.n_tiles = int(jobs);
    a.out_h = 2 * h + 1; a.out_w = 2 * w + 1;
    a.round_noise = 0; a.alpha = 0.f; a.gain = 1.f; a.clamp = -1.f;
    // Job shape (profiles/r06_convt_phase_jobs.jsonl): launches below the chip's 512 workgroup slots run one phase per workgroup; above it whole
    // position tiles when they fill their last round of slots to within 8 % (8 frames of 256^2: 4.64 rounds, +2.6 % for the pairs), phase pairs
    // otherwise (4 frames: 2.32 rounds -> 3; pairs -7 %), single phases where a phase is only a few steps long (<= 128 input channels: -10...-20 %)
    {
        const int64_t wgs = jobs * (cout / kCO), slots = 2 * kNumCU, rounds = (wgs + slots - 1) / slots;
        if (wgs < slots) a.phase_jobs = 1;
        else if ((rounds * slots - wgs) * 100 <= 8 * rounds * slots) a.phase_jobs = 0;
        else a.phase_jobs = a.cin_pad / kCK <= 2 ? 1 : 2;
    }
    if (const char* f = getenv("GNERF_CONVT_PHASE_JOBS")) { const int v = atoi(f); if (v >= 0 && v <= 2) a.phase_jobs = v; }      // (A/B runs)
    const dim3 grid((a.n_tiles + kNumXCD - 1) / kNumXCD * kNumXCD, cout / kCO, a.phase_jobs == 0 ? 1 : a.phase_jobs == 1 ? 4 : 2), block(kConvThreads);
    static PerDeviceOnce once;
    if (int rc = once.raise_lds(conv3x3_epilogue_kernel<1, false, false, false, OUT32>, what, kConvLds)) return rc;
    hipLaunchKernelGGL((conv3x3_epilogue_kernel<1, false, false, false, OUT32>), grid, block, kConvLds, as_stream(stream), a);
    return check_launch(what);
}
}  // namespace

extern "C" int gnerf_conv3x3_epilogue_nhwc(const void* x, const void* w_packed, void* y, int n, int h, int w, int cin, int cout,
                                           const float* scale, const float* noise, int round_noise, const void* bias,
                                           float alpha, float gain, float clamp, const float* next_scale, gnerf_stream_t stream) {
    return launch_conv3x3<false>("conv3x3_epilogue_nhwc", x, w_packed, y, n, h, w, cin, cout, scale, noise, round_noise, bias, alpha, gain, clamp, next_scale, stream);
}

// (ABI 11) the convolution of a block's LAST layer with the block's ToRGB in its epilogue: img[n, 3, h, w] (float32) += ToRGB(layer(x)); no y at all
extern "C" int gnerf_conv3x3_epilogue_torgb_nhwc(const void* x, const void* w_packed, int n, int h, int w, int cin,
                                                 const float* scale, const float* noise, int round_noise, const void* bias,
                                                 float alpha, float gain, float clamp,
                                                 const void* rgb_w, const float* rgb_bias, float rgb_clamp, float* img, gnerf_stream_t stream) {
    if (!img || !rgb_w) return gnerf::fail(GNERF_E_ARG, "conv3x3_epilogue_torgb_nhwc: null pointer");
    return launch_conv3x3<false>("conv3x3_epilogue_torgb_nhwc", x, w_packed, nullptr, n, h, w, cin, kCO, scale, noise, round_noise, bias, alpha, gain, clamp, nullptr, stream,
                                 rgb_w, rgb_bias, rgb_clamp, img);
}

// (ABI 10) the fp32-grade form: x3 = [hi | lo | hi] float16 from gnerf_split_f16x3_nhwc, w3 = [hi | hi | lo] packed like w_packed, float32 bias and result
extern "C" int gnerf_conv3x3_f32x3_epilogue_nhwc(const void* x3, const void* w3_packed, float* y, int n, int h, int w, int cin3, int cout,
                                                 const float* scale, const float* noise, const float* bias,
                                                 float alpha, float gain, float clamp, const float* next_scale, gnerf_stream_t stream) {
    return launch_conv3x3<true>("conv3x3_f32x3_epilogue_nhwc", x3, w3_packed, y, n, h, w, cin3, cout, scale, noise, 0, bias, alpha, gain, clamp, next_scale, stream);
}

extern "C" int gnerf_conv_transpose3x3_s2_f32x3_nhwc(const void* x3, const void* w3_phases, float* y, int n, int h, int w, int cin3, int cout, gnerf_stream_t stream) {
    return launch_conv_transpose<true>("conv_transpose3x3_s2_f32x3_nhwc", x3, w3_phases, y, n, h, w, cin3, cout, stream);
}

extern "C" int gnerf_split_f16x3_nhwc(const float* x, const float* scale, void* y, int n, int pixels, int channels, int* overflow, gnerf_stream_t stream) {
    using namespace gnerf;
    if (!x || !y) return fail(GNERF_E_ARG, "split_f16x3_nhwc: null pointer");
    if (n < 1 || pixels < 1 || channels < 8 || channels % 8) return fail(GNERF_E_UNSUPPORTED, "split_f16x3_nhwc: channels must be a positive multiple of 8 (got %d)", channels);
    if (n > 65535) return fail(GNERF_E_UNSUPPORTED, "split_f16x3_nhwc: at most 65535 images per call");
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) return fail(GNERF_E_ARG, "split_f16x3_nhwc: x and y must be 16-byte aligned");
    const int per_pixel = channels / 8;
    const int64_t total = int64_t(pixels) * per_pixel;
    // enough workgroups to fill the chip a few times over, whole pixels per grid stride (the kernel rounds the stride down to one)
    int64_t blocks = (total + 255) / 256;
    const int64_t cap = int64_t(kNumCU) * 8 / (n < 8 ? 1 : 2);
    if (blocks > cap) blocks = cap;
    if (blocks * 256 < per_pixel) blocks = (per_pixel + 255) / 256;
    hipLaunchKernelGGL(split_f16x3_kernel, dim3(unsigned(blocks), unsigned(n)), dim3(256), 0, as_stream(stream), x, scale, static_cast<_Float16*>(y), pixels, channels, overflow);
    return check_launch("split_f16x3_nhwc");
}

// y = conv_transpose2d(x, w, stride = 2) for 3x3 kernels on float16 channels_last tensors, as four phase convolutions; see include/gnerf_hip.h.
extern "C" int gnerf_conv_transpose3x3_s2_nhwc(const void* x, const void* w_phases, void* y, int n, int h, int w, int cin, int cout, gnerf_stream_t stream) {
    return launch_conv_transpose<false>("conv_transpose3x3_s2_nhwc", x, w_phases, y, n, h, w, cin, cout, stream);
}

#ifdef GNERF_CONV_STAMPS
extern "C" int gnerf_debug_conv_stamps(unsigned long long* host, int slots) {
    if (slots > 16384) slots = 16384;
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_conv_stamps), size_t(slots) * 8 * sizeof(unsigned long long)) == hipSuccess ? 0 : 1;
}
#endif
