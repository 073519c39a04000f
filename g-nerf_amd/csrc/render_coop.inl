// Cooperative render kernel: THREE WAVES PER RAY.  Included by render.hip (inside its anonymous namespace).
//
// The one-wave-per-ray kernel (render_kernel_generic) is limited by what one wave must keep: ~100 VGPRs of
// decoder weights plus 12 KiB of LDS for the colours of the 96 samples of its ray, which caps a CU at 8 waves,
// and its instruction stream repeats the same tap-address arithmetic in all 8 lanes that share a sample.
// Here a 192-lane workgroup owns one ray at a time and each wave shades every third 16-sample tile:
//   * colours stay in REGISTERS (8 per tile per lane) until the final weights are known -- no LDS spill;
//   * decoder weights live once per workgroup in LDS (padded rows, ds_read_b128 fragments right before the
//     MFMAs that use them), so a wave needs < 128 VGPRs and a CU holds 12-15 waves;
//   * bilinear tap addresses and weights are computed ONCE per (sample, plane) -- 48 lanes of a wave do the
//     3 planes of 16 samples -- and handed to the 8 lanes that read a texel through a 96-byte LDS record;
//   * plane texels are addressed as SGPR base + 32-bit VGPR offset (no 64-bit address arithmetic).
// Per-sample scalars and the merge work exactly as in the generic kernel, spread over the 192 lanes.
// Instantiated for <= 1 or 2 tiles per wave and pass (48+48 and 96+96 samples); anything else uses the
// generic kernel.

constexpr int kCoopWaves = 3;
constexpr int kCoopThreads = 64 * kCoopWaves;
constexpr int kW1Pitch = 36;        // floats per LDS row of W1 [64 x 32]
constexpr int kW2Pitch = 68;        // floats per LDS row of W2 [33 x 64]
constexpr int kTapDwords = 24;      // per sample: 3 planes x (4 byte offsets + 4 weights)
// LDS layout knobs of the forward shade tile (tools/build_variants.sh D:GNERF_TAP_STRIDE=.. / D:GNERF_STAGE_SWZ=..):
//  * record stride in dwords.  ds_write_b128 is serviced in groups of 8 consecutive lanes on 32 banks: the 8 samples of a group
//    write 16 bytes each at j * stride, conflict-free when stride / 4 is odd (28: yes, 24: 2-way).  The 8 lanes that read a record
//    share an address; a ds_read_b128 group (16 lanes, 64 banks) sees 4 distinct records: b * stride / 4 mod 16 distinct for b = 0..3.
//  * staging rows: pitch 32 with the 16-byte chunk index XORed by ((row >> 1) & 5) makes both the 8-lane row writes and the
//    (sample j, k-group g) reads of the MFMA operand conflict-free; the round-1 layout (pitch 36, no swizzle) has 2-way conflicts in
//    two of the sixteen slots of every read group.
// Round 5 defaults: stride 28, swizzled rows, the lookup window pinned plane by plane (GNERF_LOOKUP_ROLL, see coop_shade_tile).
// Counters at config 2 (profiles/r05_lds_attribution.json): SQ_LDS_BANK_CONFLICT 12.10 M -> 4.24 M per launch, of which 4.23 M are
// the scalar wave's (a build without the shade tile shows them alone): conflicts / LDS-active 0.33 -> 0.12.  The round-4 layout is
// -DGNERF_TAP_STRIDE=24 -DGNERF_STAGE_SWZ=0 -DGNERF_LOOKUP_ROLL=0.
#ifndef GNERF_TAP_STRIDE
#define GNERF_TAP_STRIDE 28
#endif
#ifndef GNERF_STAGE_SWZ
#define GNERF_STAGE_SWZ 1
#endif
#ifndef GNERF_LOOKUP_ROLL
#define GNERF_LOOKUP_ROLL 4
#endif
constexpr int kFwdTapStride = GNERF_TAP_STRIDE;
constexpr int kFwdStagePitch = GNERF_STAGE_SWZ ? 32 : kStagePitch;
static_assert(kFwdTapStride >= 24 && kFwdTapStride <= kStagePitch && kFwdTapStride % 4 == 0, "tap records live in the staging rows");
__device__ __forceinline__ int stage_swz(int row) { return GNERF_STAGE_SWZ ? ((row >> 1) & 5) : 0; }

// Decoder weights in LDS.  Two formats, BOTH shipped (template parameter MLP of the kernels; chosen per call, see
// choose_mlp in render.hip):
//  * kMlpF16x3: every fp32 weight w is stored as two halves (hi = f16(w), lo = f16(w - hi)) in MFMA fragment
//    order, and each product of the MLP is evaluated as hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_f16 with fp32
//    accumulation (the dropped lo*lo term and the split's own rounding are ~2^-21 relative, i.e. fp32-grade: the
//    renderer's pixel MSE against the reference stays ~1e-13).  24 matrix instructions x 16 cycles per 16-sample tile
//    instead of 64 x 32 cycles of v_mfma_f32_16x16x4_f32.  Only valid while features, weights and hidden activations
//    stay inside f16's range and the absolute error of the low halves (2^-25 per operand) stays negligible after the
//    decoder's own amplification -- which is what choose_mlp (render.hip) decides on the device, per call.
//  * kMlpF32: plain fp32 rows (exact fp32 products on the fp32-input MFMA): any finite input.
constexpr int kMlpAuto = 0, kMlpF16x3 = 1, kMlpF32 = 2;          // = GNERF_MLP_AUTO / GNERF_MLP_F16X3 / GNERF_MLP_F32
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
constexpr float kLog2e = 1.44269504088896341f, kLn2 = 0.693147180559945309f;

// hi/lo f16 split of eight fp32 values into two MFMA operands: hi = f16(x), lo = f16(x - hi), packed in order.
// Sixteen instructions, none of them quarter-rate: v_cvt_pk_f16_f32 per pair for hi, v_fma_mix_f32 (f16 half x -1 + fp32 value, one fused
// fp32 operation) per value for x - hi, v_cvt_pk_f16_f32 per pair for lo.  The first version formed lo directly with v_fma_mixlo/hi_f16
// (twelve instructions), but tools/probes/valu_issue_probe measures those at 8.6 SIMD cycles apiece against 4.4 for v_fma_mix_f32 and
// 4.25 for the conversion: 86 -> 69 cycles per split (the compiler's own lowering of the expression is thirty-two instructions).
// ONE asm block, ending in s_nop 1: the consumers are MFMAs, which need two wait states after a VALU write of an operand,
// and the compiler's hazard recogniser does not look inside inline asm.
#define GNERF_SPLIT_BODY                                                                                                               \
    float y0 = x[0], y1 = x[1], y2 = x[2], y3 = x[3], y4 = x[4], y5 = x[5], y6 = x[6], y7 = x[7];       /* x - hi is formed in place */   \
    asm(                                                                                                                                \
        "v_cvt_pk_f16_f32 %0, %8, %9\n\t"                                                                                               \
        "v_cvt_pk_f16_f32 %1, %10, %11\n\t"                                                                                             \
        "v_cvt_pk_f16_f32 %2, %12, %13\n\t"                                                                                             \
        "v_cvt_pk_f16_f32 %3, %14, %15\n\t"                                                                                             \
        "v_fma_mix_f32 %8, %0, -1.0, %8 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"                                                           \
        "v_fma_mix_f32 %9, %0, -1.0, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"                                                           \
        "v_fma_mix_f32 %10, %1, -1.0, %10 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"                                                         \
        "v_fma_mix_f32 %11, %1, -1.0, %11 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"                                                         \
        "v_fma_mix_f32 %12, %2, -1.0, %12 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"                                                         \
        "v_fma_mix_f32 %13, %2, -1.0, %13 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"                                                         \
        "v_fma_mix_f32 %14, %3, -1.0, %14 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"                                                         \
        "v_fma_mix_f32 %15, %3, -1.0, %15 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"                                                         \
        "v_cvt_pk_f16_f32 %4, %8, %9\n\t"                                                                                               \
        "v_cvt_pk_f16_f32 %5, %10, %11\n\t"                                                                                             \
        "v_cvt_pk_f16_f32 %6, %12, %13\n\t"                                                                                             \
        "v_cvt_pk_f16_f32 %7, %14, %15\n\t"                                                                                             \
        "s_nop 1"                                                                                                                       \
        : "=&v"(hi[0]), "=&v"(hi[1]), "=&v"(hi[2]), "=&v"(hi[3]), "=&v"(lo[0]), "=&v"(lo[1]), "=&v"(lo[2]), "=&v"(lo[3]),               \
          "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3), "+v"(y4), "+v"(y5), "+v"(y6), "+v"(y7))
__device__ __forceinline__ void split_f16x8(const float (&x)[8], unsigned (&hi)[4], unsigned (&lo)[4]) { GNERF_SPLIT_BODY; }
#undef GNERF_SPLIT_BODY
typedef unsigned u4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ h8 as_h8(u4v v) { return __builtin_bit_cast(h8, v); }
// LDS layouts that a lane (j = lane & 15, g = lane >> 4) reads 16 bytes of are kept in FRAGMENT ORDER: the chunk of lane i at
// 16 i bytes.  Measured with tools/probes/lds_probe.hip: a wave's ds_read_b128 takes 4 LDS cycles only when consecutive lanes read
// consecutive 16-byte chunks (or all read the same one); every row-per-j layout -- whatever its pitch: 16, 20, 24, 34, 36, 40, 68
// dwords were tried -- takes 8, and SQ_LDS_BANK_CONFLICT counts the other 4.  The decoder's fragments are stored that way (16 of the 22
// conflicted reads of a tile); time-neutral at config 2 (A/B on one box: 0.5845 vs 0.5852 ms per step), the LDS unit was ~44 % busy.
constexpr int kW1FragHalves = 4 * 64 * 8;   // halves of W1 hi (or lo): [m = 4 hidden blocks][lane][8 channels]
constexpr int kWeightFloatsF32 = 64 * 36 + 33 * 68;
constexpr int kWeightFloatsF16 = (2 * kW1FragHalves + 2 * 2048) / 2 + 64 + 128; // W1 hi+lo, W2 hi+lo (fragment order), density row fp32, colour biases x4
__host__ __device__ constexpr int weight_floats(int mlp) {
    return mlp == kMlpF32 ? kWeightFloatsF32 : (mlp == kMlpF16x3 ? kWeightFloatsF16 : (kWeightFloatsF32 > kWeightFloatsF16 ? kWeightFloatsF32 : kWeightFloatsF16));
}

struct CoopLds {
    float* w1; float* w2; float* b1; float* b2;
    float* b2c;      // f16x3 decoder: [32][4] the colour biases, each four times -- layer 2's accumulator initialisation is then two ds_read_b128
                     // per tile instead of a two-word read and eight v_mov_b32 (round 6)
    float* t_e; float* sig_e; float* v_e; int* rank_e; float* s_t; float* s_sig; float* w_s; float* cdf;
    float* taps;     // [waves][16][24]           wave w's records at taps + w * wave_pitch_taps
    float* stage;    // [waves][16][kStagePitch]  wave w's rows at stage + w * wave_pitch_stage
    int wave_pitch_taps, wave_pitch_stage;     // (the pipelined kernel lays the two over each other: see render_pipe_body)
    float* part;     // [waves][32] colour partial sums, then [4] ray scalars
};

__host__ __device__ inline size_t coop_lds_floats(int s_pad, int mlp) {
    return size_t(weight_floats(mlp)) + 64 + 36 + size_t(8) * s_pad +
           kCoopWaves * 16 * kFwdTapStride + kCoopWaves * 16 * kStagePitch + kCoopWaves * 32 + 4;
}

#ifdef GNERF_ABLATE_MFMA         // timing-only build: one FMA per lane instead of a matrix instruction
#define GNERF_MFMA(a, b, c) ((c) + (a) * (b))
#define GNERF_MFMA16(a, b, c) ((c) + float((a)[0]) * float((b)[0]))
#else
#define GNERF_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0)
#define GNERF_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)
#endif

// timing-only ablation hooks (GNERF_ABLATE_ACT replaces the activations by cheap linear maps)
__device__ __forceinline__ float act_softplus(float v, float x) {
#ifdef GNERF_ABLATE_ACT
    return x * 0.5f;
#else
    return v;
#endif
}
__device__ __forceinline__ float act_sigmoid_rgb(float v, float x) {
#ifdef GNERF_ABLATE_ACT
    return x * 0.25f;
#else
    return v;
#endif
}
__device__ __forceinline__ float softplus_hw(float x) {         // max(x,0) + ln2 * log2(1 + 2^(-|x| log2 e))
#ifdef GNERF_ABLATE_ACT
    return x * 0.5f;
#endif
    const float e = __builtin_amdgcn_exp2f(-fabsf(x) * 1.44269504088896341f);
    return fmaf(__builtin_amdgcn_logf(1.0f + e), 0.693147180559945309f, fmaxf(x, 0.f));
}
__device__ __forceinline__ float sigmoid_rgb_hw(float x) {      // sigmoid(x) * 1.002 - 0.001   (triplane.py:134)
#ifdef GNERF_ABLATE_ACT
    return x * 0.25f;
#endif
    const float e = __builtin_amdgcn_exp2f(x * -1.44269504088896341f);
    return fmaf(__builtin_amdgcn_rcpf(1.0f + e), 1.002f, -0.001f);
}

// Taps of one plane for one sample: 4 byte offsets into the plane (clamped, always safe to load) and the
// 4 bilinear weights (zero where the tap is outside the image; the 1/3 of the plane mean folded in).
// Byte addressing of a texel: y * row_pitch + x * tex_pitch + plane offset (Params::tex_pitch / row_pitch / plane_pitch):
// [3N,H,W,32] planes have tex_pitch 128, plane offset pl * H * W * 128; the interleaved [N,H,W,96] form (channels_last of the
// backbone's [N,96,H,W] output) has tex_pitch 384, plane offset pl * 128.  Pitches and coordinates are < 2^24: 24-bit multiplies.
#ifndef GNERF_TAPS_LEAN
#define GNERF_TAPS_LEAN 1
#endif
__device__ __forceinline__ void plane_taps(int H, int W, float u, float v, unsigned tex_pitch, unsigned row_pitch, unsigned plane_bytes_off, uint4& off, v4f& wgt) {
    float ix = ((u + 1.f) * float(W) - 1.f) * 0.5f;
    float iy = ((v + 1.f) * float(H) - 1.f) * 0.5f;
    ix = clamp_nn(ix, -1.5f, float(W) + 0.5f);           // (a NaN coordinate comes out as one of the bounds: all taps get weight 0 or
    iy = clamp_nn(iy, -1.5f, float(H) + 0.5f);           //  NaN weights, as before -- the fp32 oracle's NaN propagates either way)
    const float x0f = floorf(ix), y0f = floorf(iy);
    const float fx = ix - x0f, fy = iy - y0f;
    const int x0 = int(x0f), y0 = int(y0f), x1 = x0 + 1, y1 = y0 + 1;
#if GNERF_TAPS_LEAN
    // round 6, same values from fewer instructions: "0 <= x < W" is ONE unsigned compare (a negative index is a huge unsigned one; the
    // compiler cannot make that step itself, it does not know W > 0), and the clamp to the last texel one v_med3_i32 instead of
    // v_max_i32 + v_min_i32 (it only folds the pair when both bounds are constants): 8 compares + 4 mask ands + 8 min / max -> 4 + 0 + 4
    // per (sample, plane), all of them half-rate instructions (tools/probes/valu_issue_probe)
    auto inside = [](int i, int n) { return unsigned(i) < unsigned(n); };
    auto clamp0 = [](int i, int last) { int r; asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(i), "s"(last)); return r; };
    const float wx0 = inside(x0, W) ? 1.f - fx : 0.f, wx1 = inside(x1, W) ? fx : 0.f;
    const float wy0 = inside(y0, H) ? (1.f - fy) * (1.f / 3.f) : 0.f, wy1 = inside(y1, H) ? fy * (1.f / 3.f) : 0.f;
    const unsigned cx0 = __umul24(unsigned(clamp0(x0, W - 1)), tex_pitch), cx1 = __umul24(unsigned(clamp0(x1, W - 1)), tex_pitch);
    const unsigned cy0 = __umul24(unsigned(clamp0(y0, H - 1)), row_pitch) + plane_bytes_off;
    const unsigned cy1 = __umul24(unsigned(clamp0(y1, H - 1)), row_pitch) + plane_bytes_off;
#else
    const float wx0 = (x0 >= 0 && x0 < W) ? 1.f - fx : 0.f, wx1 = (x1 >= 0 && x1 < W) ? fx : 0.f;
    const float wy0 = (y0 >= 0 && y0 < H) ? (1.f - fy) * (1.f / 3.f) : 0.f, wy1 = (y1 >= 0 && y1 < H) ? fy * (1.f / 3.f) : 0.f;
    const unsigned cx0 = __umul24(unsigned(min(max(x0, 0), W - 1)), tex_pitch), cx1 = __umul24(unsigned(min(max(x1, 0), W - 1)), tex_pitch);
    const unsigned cy0 = __umul24(unsigned(min(max(y0, 0), H - 1)), row_pitch) + plane_bytes_off;
    const unsigned cy1 = __umul24(unsigned(min(max(y1, 0), H - 1)), row_pitch) + plane_bytes_off;
#endif
    off = make_uint4(cy0 + cx0, cy0 + cx1, cy1 + cx0, cy1 + cx1);
    wgt = (v4f){wx0 * wy0, wx1 * wy0, wx0 * wy1, wx1 * wy1};
}

// In-kernel stamps for the diagnostic build (-DGNERF_STAMPS, tools/build_variants.sh STAMPS): per-wave cycle
// totals per code segment, written to the `debug` buffer instead of the stage dump.  No stamp executes in the
// shipped library.
#ifdef GNERF_STAMPS
struct Stamps {
    unsigned long long last, acc[16];
    __device__ __forceinline__ void reset() { for (int i = 0; i < 16; i++) acc[i] = 0; last = now(); }
    static __device__ __forceinline__ unsigned long long now() {
        unsigned long long t;
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
        __builtin_amdgcn_sched_barrier(0);
        return t;
    }
    __device__ __forceinline__ void mark(int i) { const unsigned long long t = now(); acc[i] += t - last; last = t; }
};
#define GNERF_STAMP(st, i) (st).mark(i)
#else
struct Stamps { __device__ __forceinline__ void reset() {} };
#define GNERF_STAMP(st, i) ((void)0)
#endif

// Copy the decoder into LDS (all threads of the workgroup; a barrier must follow).  base = start of the weight area.
template <int MLP>
__device__ __forceinline__ void stage_decoder(CoopLds& L, float* base, const gnerf_render_params& p, int tid, int nthreads) {
    L.w1 = base;
    L.b1 = base + weight_floats(MLP);
    L.b2 = L.b1 + 64;
    if constexpr (MLP == kMlpF32) {
    L.w2 = L.w1 + 64 * kW1Pitch;
    for (int i = tid; i < 64 * 32; i += nthreads) L.w1[(i >> 5) * kW1Pitch + (i & 31)] = p.w1[i];
    for (int i = tid; i < 33 * 64; i += nthreads) L.w2[(i >> 6) * kW2Pitch + (i & 63)] = p.w2[i];
    for (int i = tid; i < 64; i += nthreads) L.b1[i] = p.b1[i];
    for (int i = tid; i < 33; i += nthreads) L.b2[i] = p.b2[i];
    } else {
    _Float16* w1h = reinterpret_cast<_Float16*>(base);                 // [hi|lo][m=4][lane = 16 g + j][8]: W1[16 m + j][8 g ..]
    _Float16* w2h = w1h + 2 * kW1FragHalves;                           // [hi|lo][n=2][s=2][lane = 16 g + j][8]
    L.w2 = base + (2 * kW1FragHalves + 2 * 2048) / 2;                  // density row W2[0][:] in fp32
    // The activations run on the hardware's base-2 exp/log, so their scale factors are folded into the weights:
    //   layer 1 produces p' = log2(e) p                      (W1, b1 scaled by log2 e)
    //   softplus becomes h' = log2(1 + 2^p') = h / ln 2       (no multiply on either side)
    //   the density row carries the ln 2 back                 (W2[0] scaled by ln 2)
    //   the colour rows produce o' = -log2(e) o = -(W2 h') - log2(e) b2, so sigmoid(o) = 1 / (1 + 2^o'):
    //   W2[1..32] is only NEGATED (exact), b2[1..32] scaled by -log2 e.
    for (int i = tid; i < 64 * 32; i += nthreads) {
        const float x = p.w1[i] * kLog2e;
        const _Float16 hi = (_Float16)x;
        const int r = i >> 5, c = i & 31;
        const int dst = (((r >> 4) * 4 + (c >> 3)) * 16 + (r & 15)) * 8 + (c & 7);          // block m, lane 16 g + j, element
        w1h[dst] = hi;
        w1h[kW1FragHalves + dst] = (_Float16)(x - (float)hi);
    }
    for (int i = tid; i < 2048; i += nthreads) {
        // fragment order of layer 2's B operand: lane (out column j, k-group g) of k-step s holds, at position jj,
        // hidden unit 32 s + 16 (jj >> 2) + 4 g + (jj & 3) -- the unit whose activation that lane group carries there
        const int jj = i & 7, g = (i >> 3) & 3, j = (i >> 5) & 15, s = (i >> 9) & 1, n = i >> 10;
        const float x = -p.w2[(1 + 16 * n + j) * 64 + 32 * s + 16 * (jj >> 2) + 4 * g + (jj & 3)];
        const _Float16 hi = (_Float16)x;
        const int dst = (((n * 2 + s) * 4 + g) * 16 + j) * 8 + jj;                           // lane 16 g + j of fragment (n, s)
        w2h[dst] = hi;
        w2h[2048 + dst] = (_Float16)(x - (float)hi);
    }
    for (int i = tid; i < 64; i += nthreads) L.w2[i] = p.w2[i] * kLn2;
    L.b2c = L.w2 + 64;
    for (int i = tid; i < 128; i += nthreads) L.b2c[i] = p.b2[1 + (i >> 2)] * -kLog2e;
    for (int i = tid; i < 64; i += nthreads) L.b1[i] = p.b1[i] * kLog2e;
    for (int i = tid; i < 33; i += nthreads) L.b2[i] = i == 0 ? p.b2[0] : p.b2[i] * -kLog2e;
    }
}

// v + (v from lane^16) + (v from lane^32) + (v from lane^48): sum over the four 16-lane rows, result in every lane.
// gfx950's v_permlane16_swap / v_permlane32_swap exchange rows between two registers; two swaps + two adds
// replace two ds_bpermute round trips through the LDS crossbar.
__device__ __forceinline__ float row_sum4(float v) {
    const unsigned u = __float_as_uint(v);
    const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);     // rows {0,1,2,3}x{0,1,2,3} -> (r0,r0,r2,r2), (r1,r1,r3,r3)
    const float s = __uint_as_float(a[0]) + __uint_as_float(a[1]);           // r0+r1 in rows 0,1 ; r2+r3 in rows 2,3
    const unsigned us = __float_as_uint(s);
    const auto b = __builtin_amdgcn_permlane32_swap(us, us, false, false);   // (lo,lo), (hi,hi)
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// Sum over the 16 lanes of a DPP row; valid in lanes with (lane & 15) == 15.
__device__ __forceinline__ float row_total(float v) {
    v += dpp_mov<0x111, 0xf>(0.f, v);
    v += dpp_mov<0x112, 0xf>(0.f, v);
    v += dpp_mov<0x114, 0xf>(0.f, v);
    v += dpp_mov<0x118, 0xf>(0.f, v);
    return v;
}

// Round 6: a lane of the tap set-up (sample j, plane pl = lane >> 4) needs two coordinates of the point, u = (ou + t du) s and
// v = (ov + t dv) s with (u, v) = (x, y), (x, z), (z, x) for planes 0, 1, 2 (renderer.py:23-53).  It used to form all three of
// x, y, z and pick two with three v_cndmask per tile; now the ray arrives as the lane's own (ou, du, ov, dv) -- the cooperative kernel
// selects once per ray, the pipelined kernels' scalar wave parks the twelve values per plane in the slot and a lane reads its four in
// one ds_read_b128 (render_pipe.inl: kMiscUV).  Same operations on the same operands: bit-identical.
struct CoopRay {
    const char* planes_item;    // uniform
    const float* sig_noise = nullptr;   // uniform: this ray's and this pass's density noise (gnerf_render_params.sigma_noise_*), or null
    float ou, du, ov, dv;       // per lane (by plane)
    __device__ __forceinline__ void set(float ox, float oy, float oz, float dx, float dy, float dz, int lane) {
        const int pl = lane >> 4;
        ou = pl == 2 ? oz : ox; du = pl == 2 ? dz : dx;
        ov = pl == 0 ? oy : (pl == 1 ? oz : ox); dv = pl == 0 ? dy : (pl == 1 ? dz : dx);
    }
};

// Shade one 16-sample tile: depths t_list[16*tile ...] (clamped to count-1) -> density into sig_list (if
// active) and this lane's 8 colour values (channel lane&15 of block n, samples 4*(lane>>4)..+3) into col.
// The tap records and the staging rows are private to the calling wave, so the two hand-offs inside need
// only wave-level ordering: BLOCK_SYNC=false uses lds_wave_sync() (s_waitcnt, no s_barrier) and the call may
// then sit in wave-divergent control flow; BLOCK_SYNC=true keeps workgroup barriers (all waves must call).
__device__ __forceinline__ void lds_wave_sync() {
    // LDS instructions of one wave execute in order, so a hand-off between lanes of the SAME wave only needs the
    // writes to have been issued before the reads: wait for this wave's LDS queue and stop the compiler from moving
    // memory operations across.  Deliberately NOT a fence: a workgroup-scope release would also wait for every global
    // load/store in flight (vmcnt(0)), serialising the scalar wave's prefetches and output stores.
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}
template <bool BLOCK_SYNC, int MLP>
__device__ __forceinline__ void coop_shade_tile(const Params& P, const CoopLds& L, const CoopRay& R, const float* t_list, int count,
                                                int tile, bool active, float* sig_list, int lane, int wv, v4f (&col)[2], Stamps& st, bool sp_direct = false,
                                                bool have_depth = false, float depth_in = 0.f) {
#ifdef GNERF_ABLATE_SHADE       // timing-only build: no lookups, no MLP
    if (active && lane < 16 && 16 * tile + lane < count) sig_list[16 * tile + lane] = t_list[16 * tile + lane] - 2.7f;
    col[0] = (v4f){0.1f, 0.2f, 0.3f, 0.4f}; col[1] = col[0];
    return;
#endif
    const int H = P.p.plane_h, W = P.p.plane_w;
    float* taps = L.taps + wv * L.wave_pitch_taps;
    float* stage = L.stage + wv * L.wave_pitch_stage;
    // ---- tap setup: lane (sample j, plane pl) for lanes 0..47
    if (lane < 48) {
        const int j = lane & 15, pl = lane >> 4;
        const int idx = min(16 * tile + j, count - 1);
        const float depth = have_depth ? depth_in : t_list[idx];       // (the pipelined kernels fetch it with the ray's parameters, one round trip)
        // plane 0 (x,y), 1 (x,z), 2 (z,x): see CoopRay.  The roundings are the ones this code has always had -- o + t d as one fused
        // operation, the box scale as a multiply of its own -- pinned: with the per-plane select gone, fp contraction would otherwise
        // merge the scale into plane_taps' `u + 1` (an empty asm is opaque to it and costs nothing)
        float u = __builtin_fmaf(depth, R.du, R.ou) * P.box_scale;
        float v = __builtin_fmaf(depth, R.dv, R.ov) * P.box_scale;
        asm("" : "+v"(u), "+v"(v));
        uint4 off; v4f wgt;
        plane_taps(H, W, u, v, P.tex_pitch, P.row_pitch, unsigned(pl) * P.plane_pitch, off, wgt);
        float* rec = taps + j * kFwdTapStride + pl * 8;
        *reinterpret_cast<uint4*>(rec) = off;
        *reinterpret_cast<v4f*>(rec + 4) = wgt;
    }
    if (BLOCK_SYNC) __syncthreads(); else lds_wave_sync();
    GNERF_STAMP(st, 1);         // tap setup
    // ---- lookup: 8 lanes per texel, 8 samples per step
    const int b = lane >> 3, cq16 = (lane & 7) * 16;
    // The tile's 24 texel loads (2 steps x 3 planes x 4 taps) run as a ROLLING window of three 4-load units (48 VGPRs in
    // flight): as soon as a plane of step 0 has been blended, the same plane of step 1 is issued, so the second step's
    // round trip overlaps the first instead of following it.  Measured alternatives (tools/ablate.py, config 2): all 24
    // in flight (96 VGPRs) halves the lookup time but costs a wave per SIMD and is slower overall; one step after the
    // other (the previous default) exposes two full round trips.
    uint4 off[2][3];
    v4f wgt[2][3], tex[2][3][4];
    auto read_records = [&](int a) {
        const float* rec = taps + (8 * a + b) * kFwdTapStride;
#pragma unroll
        for (int pl = 0; pl < 3; pl++) {
            off[a][pl] = *reinterpret_cast<const uint4*>(rec + pl * 8);
            wgt[a][pl] = *reinterpret_cast<const v4f*>(rec + pl * 8 + 4);
        }
    };
    auto issue = [&](int a, int pl) {
#ifdef GNERF_ABLATE_GATHER      // timing-only build: no texel loads (outputs are wrong)
        tex[a][pl][0] = (v4f){float(off[a][pl].x + cq16), 1.f, 2.f, 3.f}; tex[a][pl][1] = (v4f){float(off[a][pl].y), 1.f, 2.f, 3.f};
        tex[a][pl][2] = (v4f){float(off[a][pl].z), 1.f, 2.f, 3.f};        tex[a][pl][3] = (v4f){float(off[a][pl].w), 1.f, 2.f, 3.f};
#else
        tex[a][pl][0] = *reinterpret_cast<const v4f*>(R.planes_item + (off[a][pl].x + cq16));
        tex[a][pl][1] = *reinterpret_cast<const v4f*>(R.planes_item + (off[a][pl].y + cq16));
        tex[a][pl][2] = *reinterpret_cast<const v4f*>(R.planes_item + (off[a][pl].z + cq16));
        tex[a][pl][3] = *reinterpret_cast<const v4f*>(R.planes_item + (off[a][pl].w + cq16));
#endif
    };
    // One fused multiply-add per tap, chained through the sample's 12 taps (the first tap of the first plane starts the accumulator).
    // History: round 1 had this form and dropped it -- with hipcc 7.2's code for it (v_pk_fma_f32 chains whose weight operand is the
    // high register of a pair, selected on src1) lanes 48-63 of ~3 % of the rays differed from run to run; round 3's inline-asm op_sel
    // broadcast reproduced that.  Round 4 found the mechanism (csrc/pk_opsel_fixup.py: that operand form reads 0.0 in lanes 48-63 now and
    // then while another wave of the SIMD runs v_mfma_f32_16x16x32_f16); with the build exchanging the sources of every such
    // instruction (2 550 of them in this form) the chain is bit-stable (tools/determinism.py: 0 of 10 x 65 536 rays) and 1 % faster
    // than four products summed per plane (24 instead of 28 packed instructions per sample pair).
    auto blend = [&](int a, int pl, v4f& acc) {
        auto bc = [](float w) { return (v4f){w, w, w, w}; };
        acc = pl == 0 ? tex[a][pl][0] * wgt[a][pl][0] : __builtin_elementwise_fma(tex[a][pl][0], bc(wgt[a][pl][0]), acc);
        acc = __builtin_elementwise_fma(tex[a][pl][1], bc(wgt[a][pl][1]), acc);
        acc = __builtin_elementwise_fma(tex[a][pl][2], bc(wgt[a][pl][2]), acc);
        acc = __builtin_elementwise_fma(tex[a][pl][3], bc(wgt[a][pl][3]), acc);
    };
    (void)read_records; (void)issue; (void)blend;       // (the pinned orders below fetch records where they are needed)
    v4f acc0, acc1;
    float* const row0 = stage + b * kFwdStagePitch + (((lane & 7) ^ stage_swz(b)) * 4);
    float* const row1 = stage + (8 + b) * kFwdStagePitch + (((lane & 7) ^ stage_swz(8 + b)) * 4);
#if GNERF_LOOKUP_ROLL == 0
    read_records(0);
    issue(0, 0); issue(0, 1); issue(0, 2);
    read_records(1);
    __builtin_amdgcn_sched_barrier(0);
    blend(0, 0, acc0); issue(1, 0);
    __builtin_amdgcn_sched_barrier(0);
    blend(0, 1, acc0); issue(1, 1);
    __builtin_amdgcn_sched_barrier(0);
    blend(0, 2, acc0); issue(1, 2);
    __builtin_amdgcn_sched_barrier(0);
    // (the staging rows stay [sample][kStagePitch]: 2-way conflicts on the writes and on the reads; fragment order for the reads makes
    //  the writes 8-way -- 8 lanes of a sample land on one bank group -- and was 3 % slower overall)
    *reinterpret_cast<v4f*>(row0) = acc0;
    blend(1, 0, acc1); blend(1, 1, acc1); blend(1, 2, acc1);
    *reinterpret_cast<v4f*>(row1) = acc1;
#else
    // The rolling window, PINNED.  __builtin_amdgcn_sched_barrier only stops the machine scheduler: the blends are pure arithmetic
    // with no chain to it, and instruction selection linearised all of step 0's blends in front of the first barrier -- the emitted
    // stream was 12 loads, 12 waits + blends, 12 loads, 12 waits + blends: two exposed round trips per tile.  An empty asm that takes
    // the accumulator in and out and clobbers memory has a data edge to the blend before and after it and an order edge to the loads.
    auto pin = [](v4f& a) { asm volatile("" : "+v"(a) :: "memory"); };
    // Records are fetched from LDS where they are needed (offsets right before a plane's loads leave, step 1's weights right before
    // its blend): with both steps' records held in registers (48) next to the window (48) and the colours (32) the pinned order spills.
    const float* const rec0 = taps + b * kFwdTapStride;
    const float* const rec1 = taps + (8 + b) * kFwdTapStride;
    auto bc = [](float w) { return (v4f){w, w, w, w}; };
    auto load4 = [&](v4f (&t)[4], uint4 o) {
#ifdef GNERF_ABLATE_GATHER      // timing-only build: no texel loads (outputs are wrong): what a perfect load latency could return
        t[0] = (v4f){float(o.x + cq16), 1.f, 2.f, 3.f}; t[1] = (v4f){float(o.y), 1.f, 2.f, 3.f};
        t[2] = (v4f){float(o.z), 1.f, 2.f, 3.f};        t[3] = (v4f){float(o.w), 1.f, 2.f, 3.f};
#else
        t[0] = *reinterpret_cast<const v4f*>(R.planes_item + (o.x + cq16));
        t[1] = *reinterpret_cast<const v4f*>(R.planes_item + (o.y + cq16));
        t[2] = *reinterpret_cast<const v4f*>(R.planes_item + (o.z + cq16));
        t[3] = *reinterpret_cast<const v4f*>(R.planes_item + (o.w + cq16));
#endif
    };
    v4f win[3][4], w0[3];
#pragma unroll
    for (int pl = 0; pl < 3; pl++) {
        load4(win[pl], *reinterpret_cast<const uint4*>(rec0 + pl * 8));
        w0[pl] = *reinterpret_cast<const v4f*>(rec0 + pl * 8 + 4);
    }
#pragma unroll
    for (int pl = 0; pl < 3; pl++) {
        const uint4 o1 = *reinterpret_cast<const uint4*>(rec1 + pl * 8);
#if GNERF_LOOKUP_ROLL == 4          // a plane's four taps at a time
        acc0 = pl == 0 ? win[pl][0] * w0[pl][0] : __builtin_elementwise_fma(win[pl][0], bc(w0[pl][0]), acc0);
        acc0 = __builtin_elementwise_fma(win[pl][1], bc(w0[pl][1]), acc0);
        acc0 = __builtin_elementwise_fma(win[pl][2], bc(w0[pl][2]), acc0);
        acc0 = __builtin_elementwise_fma(win[pl][3], bc(w0[pl][3]), acc0);
        pin(acc0);
        load4(win[pl], o1);
        pin(acc0);
#else                               // tap by tap: a load of step 1 leaves as soon as the same tap of step 0 has been consumed
#pragma unroll
        for (int t = 0; t < 4; t++) {
            acc0 = (pl == 0 && t == 0) ? win[pl][t] * w0[pl][t] : __builtin_elementwise_fma(win[pl][t], bc(w0[pl][t]), acc0);
            pin(acc0);
            const unsigned ot = t == 0 ? o1.x : (t == 1 ? o1.y : (t == 2 ? o1.z : o1.w));
            win[pl][t] = *reinterpret_cast<const v4f*>(R.planes_item + (ot + cq16));
            pin(acc0);
        }
#endif
    }
    // (the records share their LDS with the staging rows: step 1's weights leave it BEFORE the first row is written over them)
    v4f w1[3];
#pragma unroll
    for (int pl = 0; pl < 3; pl++) w1[pl] = *reinterpret_cast<const v4f*>(rec1 + pl * 8 + 4);
    pin(acc0);
    *reinterpret_cast<v4f*>(row0) = acc0;
#pragma unroll
    for (int pl = 0; pl < 3; pl++) {
        acc1 = pl == 0 ? win[pl][0] * w1[pl][0] : __builtin_elementwise_fma(win[pl][0], bc(w1[pl][0]), acc1);
        acc1 = __builtin_elementwise_fma(win[pl][1], bc(w1[pl][1]), acc1);
        acc1 = __builtin_elementwise_fma(win[pl][2], bc(w1[pl][2]), acc1);
        acc1 = __builtin_elementwise_fma(win[pl][3], bc(w1[pl][3]), acc1);
    }
    *reinterpret_cast<v4f*>(row1) = acc1;
#endif
    if (BLOCK_SYNC) __syncthreads(); else lds_wave_sync();
    GNERF_STAMP(st, 2);         // lookups (tap records, texel loads, blend, staging)
    const int j = lane & 15, g = lane >> 4;
    const v4f f_lo = *reinterpret_cast<const v4f*>(stage + j * kFwdStagePitch + (((2 * g) ^ stage_swz(j)) * 4));
    const v4f f_hi = *reinterpret_cast<const v4f*>(stage + j * kFwdStagePitch + (((2 * g + 1) ^ stage_swz(j)) * 4));
    const float f[8] = {f_lo[0], f_lo[1], f_lo[2], f_lo[3], f_hi[0], f_hi[1], f_hi[2], f_hi[3]};
    float sig = 0.f;
    v4f o[2];
    if constexpr (MLP == kMlpF32) {
    // ---- the MLP as one software-pipelined MFMA stream (64 matrix instructions back to back):
    //   L1(0) | L1(1)+SP(0) | L1(2)+SP(1) | L1(3)+SP(2) | L2(0)+SP(3) | L2(1) | L2(2) | L2(3) | sigmoid
    // L1(m): the 8 MFMAs of hidden block m (W1 rows 16m + j, columns 8g..8g+7; bias preloaded in the accumulator).
    // SP(m): softplus of block m's 4 accumulator values, issued in 4-wide groups (exp x4, log x4, fma x4) in the
    // shadow of the MFMAs of the next block -- four independent chains hide the transcendental latency, and the
    // groups sit between MFMAs in program order because a wave issues in order.
    // L2(m): 8 MFMAs consuming block m's activations (W2 rows 1 + 16n + j, columns 16m + 4g..+3); the density row is a
    // 4-term FMA per block.  GNERF_SCHED() pins the interleave against the compiler's scheduler.
    // weight fragments are fetched from LDS one block ahead of their MFMAs (a few live registers, latency covered)
    v4f h[4], a_lo[4], a_hi[4], ws[4], w0[4], w1[4];
    auto load_w1 = [&](int m) {
        a_lo[m] = *reinterpret_cast<const v4f*>(L.w1 + (16 * m + j) * kW1Pitch + 8 * g);
        a_hi[m] = *reinterpret_cast<const v4f*>(L.w1 + (16 * m + j) * kW1Pitch + 8 * g + 4);
    };
    auto load_w2 = [&](int m) {
        ws[m] = *reinterpret_cast<const v4f*>(L.w2 + 16 * m + 4 * g);
        w0[m] = *reinterpret_cast<const v4f*>(L.w2 + (1 + j) * kW2Pitch + 16 * m + 4 * g);
        w1[m] = *reinterpret_cast<const v4f*>(L.w2 + (17 + j) * kW2Pitch + 16 * m + 4 * g);
    };
#pragma unroll
    for (int m = 0; m < 4; m++) h[m] = *reinterpret_cast<const v4f*>(L.b1 + 16 * m + 4 * g);
    load_w1(0);
    load_w1(1);
    const float bc0 = L.b2[1 + j], bc1 = L.b2[17 + j];
    o[0] = (v4f){bc0, bc0, bc0, bc0}; o[1] = (v4f){bc1, bc1, bc1, bc1};
    v4f e, hv[4];
    auto sp_exp = [&](int m) {
#pragma unroll
        for (int r = 0; r < 4; r++) e[r] = __builtin_amdgcn_exp2f(-fabsf(h[m][r]) * 1.44269504088896341f);
    };
    auto sp_log = [&]() {
#pragma unroll
        for (int r = 0; r < 4; r++) e[r] = __builtin_amdgcn_logf(1.0f + e[r]);
    };
    auto sp_fin = [&](int m) {
#pragma unroll
        for (int r = 0; r < 4; r++) hv[m][r] = act_softplus(fmaf(e[r], 0.693147180559945309f, fmaxf(h[m][r], 0.f)), h[m][r]);
    };
#define GNERF_SCHED() __builtin_amdgcn_sched_barrier(0)
#pragma unroll
    for (int s = 0; s < 4; s++) h[0] = GNERF_MFMA(a_lo[0][s], f[s], h[0]);
#pragma unroll
    for (int s = 0; s < 4; s++) h[0] = GNERF_MFMA(a_hi[0][s], f[4 + s], h[0]);
#pragma unroll
    for (int m = 1; m < 4; m++) {
        if (m < 3) load_w1(m + 1); else load_w2(0);
        h[m] = GNERF_MFMA(a_lo[m][0], f[0], h[m]); h[m] = GNERF_MFMA(a_lo[m][1], f[1], h[m]);
        GNERF_SCHED(); sp_exp(m - 1); GNERF_SCHED();
        h[m] = GNERF_MFMA(a_lo[m][2], f[2], h[m]); h[m] = GNERF_MFMA(a_lo[m][3], f[3], h[m]);
        GNERF_SCHED(); sp_log(); GNERF_SCHED();
        h[m] = GNERF_MFMA(a_hi[m][0], f[4], h[m]); h[m] = GNERF_MFMA(a_hi[m][1], f[5], h[m]);
        GNERF_SCHED(); sp_fin(m - 1); GNERF_SCHED();
        h[m] = GNERF_MFMA(a_hi[m][2], f[6], h[m]); h[m] = GNERF_MFMA(a_hi[m][3], f[7], h[m]);
    }
#ifdef GNERF_STAMPS
    asm volatile("" :: "v"(h[3][0]));      // wait for layer 1 before stamping
#endif
    GNERF_STAMP(st, 3);         // layer 1 (+ softplus of blocks 0..2)
#pragma unroll
    for (int m = 0; m < 4; m++) {
        if (m < 3) load_w2(m + 1);
        o[0] = GNERF_MFMA(hv[m][0], w0[m][0], o[0]); o[1] = GNERF_MFMA(hv[m][0], w1[m][0], o[1]);
        if (m == 0) { GNERF_SCHED(); sp_exp(3); GNERF_SCHED(); }
        o[0] = GNERF_MFMA(hv[m][1], w0[m][1], o[0]); o[1] = GNERF_MFMA(hv[m][1], w1[m][1], o[1]);
        if (m == 0) { GNERF_SCHED(); sp_log(); GNERF_SCHED(); }
        o[0] = GNERF_MFMA(hv[m][2], w0[m][2], o[0]); o[1] = GNERF_MFMA(hv[m][2], w1[m][2], o[1]);
        if (m == 0) { GNERF_SCHED(); sp_fin(3); GNERF_SCHED(); }
        o[0] = GNERF_MFMA(hv[m][3], w0[m][3], o[0]); o[1] = GNERF_MFMA(hv[m][3], w1[m][3], o[1]);
#pragma unroll
        for (int r = 0; r < 4; r++) sig = fmaf(ws[m][r], hv[m][r], sig);
    }
#undef GNERF_SCHED
    } else {
    // ---- the MLP on v_mfma_f32_16x16x32_f16, every product as hi*hi + hi*lo + lo*hi with fp32 accumulation.
    // Layer 1: H^T block m [16 hidden x 16 samples] = W1 block [16 x 32] . X^T [32 x 16]: ONE k-step (K = 32 channels);
    // A = this lane's 8 weights W1[16m + j][8g..8g+7], B = its 8 staged features (channels 8g..8g+7 of sample j).
    const _Float16* w1h = reinterpret_cast<const _Float16*>(L.w1);
    const _Float16* w2h = w1h + 2 * kW1FragHalves;
    unsigned fh_u[4], fl_u[4];
    split_f16x8(f, fh_u, fl_u);
    const h8 fh = as_h8((u4v){fh_u[0], fh_u[1], fh_u[2], fh_u[3]}), fl = as_h8((u4v){fl_u[0], fl_u[1], fl_u[2], fl_u[3]});
    v4f h[4];
    h8 a_hi[4], a_lo[4];
#pragma unroll
    for (int m = 0; m < 4; m++) {
        h[m] = *reinterpret_cast<const v4f*>(L.b1 + 16 * m + 4 * g);
        a_hi[m] = *reinterpret_cast<const h8*>(w1h + (m * 64 + lane) * 8);
        a_lo[m] = *reinterpret_cast<const h8*>(w1h + kW1FragHalves + (m * 64 + lane) * 8);
    }
#pragma unroll
    for (int m = 0; m < 4; m++) h[m] = GNERF_MFMA16(a_hi[m], fh, h[m]);
#pragma unroll
    for (int m = 0; m < 4; m++) h[m] = GNERF_MFMA16(a_hi[m], fl, h[m]);
#pragma unroll
    for (int m = 0; m < 4; m++) h[m] = GNERF_MFMA16(a_lo[m], fh, h[m]);
    // layer-2 weight fragments (B operand) for n = 0,1 and k-step s = 0,1; fetched only now (the scheduling barrier
    // keeps the compiler from hoisting these 32 registers above layer 1, where they would force spills)
    __builtin_amdgcn_sched_barrier(0);
    h8 w_hi[2][2], w_lo[2][2];
#pragma unroll
    for (int n = 0; n < 2; n++) {
#pragma unroll
        for (int s = 0; s < 2; s++) {
            w_hi[n][s] = *reinterpret_cast<const h8*>(w2h + ((n * 2 + s) * 64 + lane) * 8);
            w_lo[n][s] = *reinterpret_cast<const h8*>(w2h + 2048 + ((n * 2 + s) * 64 + lane) * 8);
        }
    }
#ifdef GNERF_STAMPS
    asm volatile("" :: "v"(h[3][0]));
#endif
    GNERF_STAMP(st, 3);         // layer 1
    // softplus in 4-wide groups (four independent exp2/log2 chains), density row in fp32 on the VALU
    v4f hv[4];
#pragma unroll
    for (int m = 0; m < 4; m++) {
        // h' = log2(1 + 2^p') = max(p', 0) + l with l = log2(1 + 2^-|p'|) = max(p' + l, l): no overflow for any p' (so no clamp in
        // front of the exp2), and both operands of the max are arithmetic results (no canonicalising v_max in IEEE mode).  Per value:
        // v_exp_f32 (-|p'| as source modifiers), v_add_f32, v_log_f32, v_add_f32, v_max_f32 = 2 quarter-rate + 2 full-rate + 1
        // half-rate instruction (tools/probes/valu_issue_probe: 8.4 + 2.3 + 8.3 + 2.3 + 4.5 SIMD cycles) where
        // min / [canonicalise] / exp2 / add / log2 / max was 2 quarter-rate + 1 full-rate + 3 half-rate.
        // When the decoder's norms bound every |p'| below exp2's overflow (choose_mlp: sp_direct, wave-uniform) the short form
        // log2(1 + 2^p') does: exp2, add, log2 -- 16 v_add and 16 half-rate v_max fewer per tile, 3.3 % of the kernel at config 2.
        v4f e;
        if (sp_direct) {
#pragma unroll
            for (int r = 0; r < 4; r++) e[r] = __builtin_amdgcn_exp2f(h[m][r]);
#pragma unroll
            for (int r = 0; r < 4; r++) hv[m][r] = act_softplus(__builtin_amdgcn_logf(1.0f + e[r]), h[m][r]);
        } else {
#pragma unroll
            for (int r = 0; r < 4; r++) e[r] = __builtin_amdgcn_exp2f(-fabsf(h[m][r]));
#pragma unroll
            for (int r = 0; r < 4; r++) e[r] = __builtin_amdgcn_logf(1.0f + e[r]);
#pragma unroll
            for (int r = 0; r < 4; r++) hv[m][r] = act_softplus(fmaxf(h[m][r] + e[r], e[r]), h[m][r]);
        }
        const v4f ws = *reinterpret_cast<const v4f*>(L.w2 + 16 * m + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; r++) sig = fmaf(ws[r], hv[m][r], sig);
    }
    // Layer 2: O [16 samples x 16 outs] = H [16 x 64] . W2^T: two k-steps of 32; this lane contributes, at k-step s,
    // its activations of blocks 2s and 2s+1 (weights were stored in the matching order by stage_decoder).
    o[0] = *reinterpret_cast<const v4f*>(L.b2c + 4 * j); o[1] = *reinterpret_cast<const v4f*>(L.b2c + 64 + 4 * j);
    h8 x_hi[2], x_lo[2];
#pragma unroll
    for (int s = 0; s < 2; s++) {
        unsigned xh[4], xl[4];
        const float xs[8] = {hv[2 * s][0], hv[2 * s][1], hv[2 * s][2], hv[2 * s][3], hv[2 * s + 1][0], hv[2 * s + 1][1], hv[2 * s + 1][2], hv[2 * s + 1][3]};
        split_f16x8(xs, xh, xl);
        x_hi[s] = as_h8((u4v){xh[0], xh[1], xh[2], xh[3]});
        x_lo[s] = as_h8((u4v){xl[0], xl[1], xl[2], xl[3]});
    }
#pragma unroll
    for (int s = 0; s < 2; s++) {
#pragma unroll
        for (int n = 0; n < 2; n++) o[n] = GNERF_MFMA16(x_hi[s], w_hi[n][s], o[n]);
    }
#pragma unroll
    for (int s = 0; s < 2; s++) {
#pragma unroll
        for (int n = 0; n < 2; n++) o[n] = GNERF_MFMA16(x_hi[s], w_lo[n][s], o[n]);
    }
#pragma unroll
    for (int s = 0; s < 2; s++) {
#pragma unroll
        for (int n = 0; n < 2; n++) o[n] = GNERF_MFMA16(x_lo[s], w_hi[n][s], o[n]);
    }
    }
    sig = row_sum4(sig) + L.b2[0];
    if (active && g == 0 && 16 * tile + j < count) {
        if (R.sig_noise) sig += R.sig_noise[16 * tile + j];           // renderer.py:146-147 (wave-uniform; never set in the compile-time-count instantiations)
        sig_list[16 * tile + j] = sig;
    }
    // rgb = sigmoid(o) * 1.002 - 0.001 (triplane.py:134), again in 4-wide groups
#pragma unroll
    for (int n = 0; n < 2; n++) {
        v4f t;
        if constexpr (MLP == kMlpF32) {
#pragma unroll
            for (int r = 0; r < 4; r++) t[r] = __builtin_amdgcn_exp2f(o[n][r] * -1.44269504088896341f);
        } else {
#pragma unroll
            for (int r = 0; r < 4; r++) t[r] = __builtin_amdgcn_exp2f(o[n][r]);        // o already carries the -log2(e)
        }
#pragma unroll
        for (int r = 0; r < 4; r++) t[r] = __builtin_amdgcn_rcpf(1.0f + t[r]);
#pragma unroll
        for (int r = 0; r < 4; r++) col[n][r] = act_sigmoid_rgb(fmaf(t[r], 1.002f, -0.001f), o[n][r]);
    }
#ifdef GNERF_STAMPS
    asm volatile("" :: "v"(col[0][0]), "v"(col[1][3]));
#endif
    GNERF_STAMP(st, 4);         // activations + layer 2
}

template <int TC1, int TF1, int MLP>
__device__ __forceinline__ void render_coop_body(const Params& P, float* smem, bool sp_direct = false) {
    const gnerf_render_params& p = P.p;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int S = p.depth_resolution, F = p.depth_resolution_importance;
    const int s_pad = 16 * (P.tiles_c + P.tiles_f);
    CoopLds L;
    L.t_e = smem + weight_floats(MLP) + 64 + 36;
    L.sig_e = L.t_e + s_pad;
    L.v_e = L.sig_e + s_pad;
    L.rank_e = reinterpret_cast<int*>(L.v_e + s_pad);
    L.s_t = L.v_e + 2 * s_pad;
    L.s_sig = L.s_t + s_pad;
    L.w_s = L.s_sig + s_pad;
    L.cdf = L.w_s + s_pad;
    L.taps = L.cdf + s_pad;
    L.stage = L.taps + kCoopWaves * 16 * kFwdTapStride;
    L.wave_pitch_taps = 16 * kFwdTapStride; L.wave_pitch_stage = 16 * kStagePitch;
    L.part = L.stage + kCoopWaves * 16 * kStagePitch;

    const int n_groups = P.n_tiles << P.split_shift;
    const int per_xcd = (n_groups + kNumXCD - 1) / kNumXCD;
    const int group = (blockIdx.x % kNumXCD) * per_xcd + blockIdx.x / kNumXCD;
    const int tile_id = group >> P.split_shift;
    const int rr_count = group < n_groups ? (kRaysPerWave >> P.split_shift) : 0;        // surplus workgroups of the rounded-up grid only report in
    const int rr_first = (group & ((1 << P.split_shift) - 1)) * (kRaysPerWave >> P.split_shift);

    // decoder -> LDS (padded rows), once per workgroup
    stage_decoder<MLP>(L, smem, p, tid, kCoopThreads);

    const int fine_e0 = 16 * P.tiles_c;
    const int n_all = S + F;
    DepthRange range;
    Stamps st;
    st.reset();

    for (int rr = rr_first; rr < rr_first + rr_count; rr++) {
        int64_t ray;
        if (P.tiles_per_item > 0) {
            const int item = tile_id / P.tiles_per_item, tt = tile_id % P.tiles_per_item;
            const int tx = tt / P.tiles_y, ty = tt % P.tiles_y;
            ray = int64_t(item) * p.rays_per_item + int64_t(ty * 4 + (rr >> 2)) * p.image_width + tx * 4 + (rr & 3);
        } else {
            ray = int64_t(tile_id) * kRaysPerWave + rr;
            if (ray >= P.total_rays) break;
        }
        const int item = int(ray / p.rays_per_item);
        CoopRay R;
        R.planes_item = reinterpret_cast<const char*>(p.planes_nhwc) + int64_t(item) * P.item_bytes;
        R.set(p.ray_origins[ray * 3 + 0], p.ray_origins[ray * 3 + 1], p.ray_origins[ray * 3 + 2],
              p.ray_dirs[ray * 3 + 0], p.ray_dirs[ray * 3 + 1], p.ray_dirs[ray * 3 + 2], lane);
        float* dbg = p.debug ? p.debug + ray * GNERF_DEBUG_SLOTS * n_all : nullptr;

        // ---- stratified depth proposals (renderer.py:169-192)
        for (int k = tid; k < S; k += kCoopThreads) {
            const float u = p.noise_coarse[ray * S + k];
            float d;
            if (p.disparity_space_sampling) {
                const float step = 1.0f / float(S - 1);
                const float lin = (k < S / 2) ? __fmul_rn(step, float(k)) : __fsub_rn(1.0f, __fmul_rn(step, float(S - 1 - k)));
                const float q = __fadd_rn(lin, __fmul_rn(u, P.disp_delta));
                d = __fdiv_rn(1.0f, __fadd_rn(__fmul_rn(P.inv_start, __fsub_rn(1.0f, q)), __fmul_rn(P.inv_end, q)));
            } else if (p.ray_start_per_ray) {
                const float rs = p.ray_start_per_ray[ray], re = p.ray_end_per_ray[ray];
                const float span = __fsub_rn(re, rs);
                const float lin = __fadd_rn(rs, __fmul_rn(__fdiv_rn(float(k), float(S - 1)), span));
                d = __fadd_rn(lin, __fmul_rn(u, __fdiv_rn(span, float(S - 1))));
            } else {
                const float step = __fdiv_rn(__fsub_rn(p.ray_end, p.ray_start), float(S - 1));
                const float lin = (k < S / 2) ? __fadd_rn(p.ray_start, __fmul_rn(step, float(k)))
                                              : __fsub_rn(p.ray_end, __fmul_rn(step, float(S - 1 - k)));
                d = __fadd_rn(lin, __fmul_rn(u, P.delta));
            }
            L.t_e[k] = d;
            if (dbg) dbg[GNERF_DBG_DEPTH_COARSE * n_all + k] = d;
        }
        for (int k = tid; k < s_pad; k += kCoopThreads) {
            L.v_e[k] = 0.f;
            if ((k >= S && k < fine_e0) || k >= fine_e0 + F) L.t_e[k] = INFINITY;      // tile padding sorts last
        }
        __syncthreads();

        // ---- coarse pass: wave wv shades tiles wv, wv+3, ...
        v4f col_c[TC1][2], col_f[TF1 > 0 ? TF1 : 1][2];
#pragma unroll
        for (int i = 0; i < TC1; i++) {
            const int t = wv + kCoopWaves * i;
            R.sig_noise = p.sigma_noise_coarse ? p.sigma_noise_coarse + ray * S : nullptr;
            coop_shade_tile<true, MLP>(P, L, R, L.t_e, S, t, t < P.tiles_c, L.sig_e, lane, wv, col_c[i], st, sp_direct);
        }
        __syncthreads();
        if (dbg) for (int k = tid; k < S; k += kCoopThreads) dbg[GNERF_DBG_SIGMA_COARSE * n_all + k] = L.sig_e[k];

        float w_sum = 0.f, wt_sum = 0.f;
        if (TF1 > 0 && F > 0) {
            const int n_w = S - 3;
            if (wv == 0) {
                march(L.t_e, L.sig_e, L.w_s, S, lane, w_sum, wt_sum);
                // wave-local hand-off of w_s (LDS ops of one wave execute in order; stop the compiler reordering)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                float part = 0.f;
                for (int i = lane; i < n_w; i += 64) {
                    const float w0 = L.w_s[i], w1 = L.w_s[i + 1], w2 = L.w_s[i + 2];
                    const float pw = ((fmaxf(w0, w1) + fmaxf(w1, w2)) * 0.5f + 0.01f) + 1e-5f;
                    L.s_sig[i] = pw;
                    part += pw;
                }
                const float inv_total = __builtin_amdgcn_rcpf(wave_sum(part));
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                float carry = 0.f;
                for (int base = 0; base < n_w; base += 64) {
                    const int i = base + lane;
                    const float pdf = (i < n_w) ? L.s_sig[i] * inv_total : 0.f;
                    const float incl = wave_scan_add(pdf, lane) + carry;
                    if (i < n_w) L.cdf[i + 1] = incl;
                    carry = wave_last(incl);
                }
                if (lane == 0) L.cdf[0] = 0.f;
            }
            __syncthreads();
            if (dbg) for (int k = tid; k < S - 1; k += kCoopThreads) dbg[GNERF_DBG_WEIGHT_COARSE * n_all + k] = L.w_s[k];
            for (int i = tid; i < F; i += kCoopThreads) {
                const float u = p.noise_fine[ray * F + i];
                int lo = 0, hi = n_w + 1;
                while (lo < hi) { const int mid = (lo + hi) >> 1; if (L.cdf[mid] <= u) lo = mid + 1; else hi = mid; }
                const int below = max(lo - 1, 0), above = min(lo, n_w);
                const float cb = L.cdf[below], ca = L.cdf[above];
                const float bb = (L.t_e[below] + L.t_e[below + 1]) * 0.5f;
                const float ba = (L.t_e[above] + L.t_e[above + 1]) * 0.5f;
                float denom = ca - cb;
                if (denom < 1e-5f) denom = 1.f;
                const float d = bb + (u - cb) * __builtin_amdgcn_rcpf(denom) * (ba - bb);
                L.t_e[fine_e0 + i] = d;
                if (dbg) dbg[GNERF_DBG_DEPTH_FINE * n_all + i] = d;
            }
            __syncthreads();

            // ---- fine pass
#pragma unroll
            for (int i = 0; i < TF1; i++) {
                const int t = wv + kCoopWaves * i;
                R.sig_noise = p.sigma_noise_fine ? p.sigma_noise_fine + ray * F : nullptr;
                coop_shade_tile<true, MLP>(P, L, R, L.t_e + fine_e0, F, t, t < P.tiles_f, L.sig_e + fine_e0, lane, wv, col_f[i], st, sp_direct);
            }
            __syncthreads();
            if (dbg) for (int k = tid; k < F; k += kCoopThreads) dbg[GNERF_DBG_SIGMA_FINE * n_all + k] = L.sig_e[fine_e0 + k];

            // ---- merge by depth (renderer.py:157-167) = stable rank of every element of cat([coarse, fine]).
            // Coarse depths ascend by construction (lin_k + u*delta with u < 1) up to rounding, so a coarse sample's rank is
            // k + #(fine before it) and a fine sample's is #(coarse <= it) [binary search] + #(fine before it).
            // Wave 0 ranks the fine samples, wave 1 the coarse ones; both count over the fine keys, read four at
            // a time as LDS broadcasts.  Ties: coarse before fine, lower index first (what a stable sort gives).
#ifdef GNERF_ABLATE_RANK
            for (int q = tid; q < n_all; q += kCoopThreads) { const int e = q < S ? q : fine_e0 + (q - S); L.rank_e[e] = q; L.s_t[q] = L.t_e[e]; L.s_sig[q] = L.sig_e[e]; }
#else
            if (wv == 0) {
                for (int i = lane; i < F; i += 64) {
                    const float key = L.t_e[fine_e0 + i];
                    int cnt = 0;
                    for (int o2 = 0; o2 < 16 * P.tiles_f; o2 += 4) {
                        const v4f k4 = *reinterpret_cast<const v4f*>(L.t_e + fine_e0 + o2);
#pragma unroll
                        for (int c2 = 0; c2 < 4; c2++) cnt += (k4[c2] < key || (k4[c2] == key && o2 + c2 < i)) ? 1 : 0;
                    }
                    int lo = 0, hi = S;                                  // number of coarse depths <= key
                    while (lo < hi) { const int mid = (lo + hi) >> 1; if (L.t_e[mid] <= key) lo = mid + 1; else hi = mid; }
                    // neighbouring coarse depths can be swapped by one ulp (see below), which can put the search off by
                    // one: recount exactly in a window of four around its answer
                    const int w0 = max(lo - 2, 0), w1 = min(lo + 2, S);
                    lo = w0;
                    for (int q = w0; q < w1; q++) lo += (L.t_e[q] <= key) ? 1 : 0;
                    const int rank = cnt + lo;
                    L.rank_e[fine_e0 + i] = rank;
                    L.s_t[rank] = key;
                    L.s_sig[rank] = L.sig_e[fine_e0 + i];
                }
            } else if (wv == 1) {
                for (int k = lane; k < S; k += 64) {
                    const float key = L.t_e[k];
                    int cnt = 0;
                    for (int o2 = 0; o2 < 16 * P.tiles_f; o2 += 4) {
                        const v4f k4 = *reinterpret_cast<const v4f*>(L.t_e + fine_e0 + o2);
#pragma unroll
                        for (int c2 = 0; c2 < 4; c2++) cnt += (k4[c2] < key) ? 1 : 0;
                    }
                    // t_k = lin_k + u delta can round one ulp past t_{k+1} when u is within ~1e-5 of 1; only neighbours can
                    // swap (the grid step is ~1e5 ulps), so the coarse samples sorted before k number k, k+1 or k-1
                    if (k + 1 < S && L.t_e[k + 1] < key) cnt += 1;
                    if (k > 0 && L.t_e[k - 1] > key) cnt -= 1;
                    const int rank = k + cnt;
                    L.rank_e[k] = rank;
                    L.s_t[rank] = key;
                    L.s_sig[rank] = L.sig_e[k];
                }
            }
#endif
            __syncthreads();
            if (wv == 0) {
                march(L.s_t, L.s_sig, L.w_s, n_all, lane, w_sum, wt_sum);
                if (lane == 0) { L.part[kCoopWaves * 32] = w_sum; range.add(P, item, L.s_t[0], L.s_t[n_all - 1]); }
            }
            __syncthreads();
            for (int q = tid; q < n_all; q += kCoopThreads) {
                const int e = q < S ? q : fine_e0 + (q - S);
                const int r = L.rank_e[e];
                const float wl = r > 0 ? L.w_s[r - 1] : 0.f, wr = r < n_all - 1 ? L.w_s[r] : 0.f;
                L.v_e[e] = (wl + wr) * 0.5f;
            }
            if (dbg) {
                for (int k = tid; k < n_all; k += kCoopThreads) { dbg[GNERF_DBG_DEPTH_SORTED * n_all + k] = L.s_t[k]; dbg[GNERF_DBG_SIGMA_SORTED * n_all + k] = L.s_sig[k]; }
                for (int k = tid; k < n_all - 1; k += kCoopThreads) dbg[GNERF_DBG_WEIGHT_FINAL * n_all + k] = L.w_s[k];
            }
        } else {
            if (wv == 0) {
                march(L.t_e, L.sig_e, L.w_s, S, lane, w_sum, wt_sum);
                float mn = INFINITY, mx = -INFINITY;
                for (int k = lane; k < S; k += 64) { mn = fminf(mn, L.t_e[k]); mx = fmaxf(mx, L.t_e[k]); }
#pragma unroll
                for (int o2 = 32; o2 > 0; o2 >>= 1) { mn = fminf(mn, __shfl_xor(mn, o2)); mx = fmaxf(mx, __shfl_xor(mx, o2)); }
                if (lane == 0) { range.add(P, item, mn, mx); L.part[kCoopWaves * 32] = w_sum; }
            }
            __syncthreads();
            for (int k = tid; k < S; k += kCoopThreads) {
                const float wl = k > 0 ? L.w_s[k - 1] : 0.f, wr = k < S - 1 ? L.w_s[k] : 0.f;
                L.v_e[k] = (wl + wr) * 0.5f;
            }
            if (dbg) for (int k = tid; k < S - 1; k += kCoopThreads) dbg[GNERF_DBG_WEIGHT_FINAL * n_all + k] = L.w_s[k];
        }
        __syncthreads();

        // ---- colours: each wave sums its own tiles, partials meet in LDS
        const int j = lane & 15, g = lane >> 4;
        float acc[2] = {0.f, 0.f};
#pragma unroll
        for (int i = 0; i < TC1; i++) {
            const int t = wv + kCoopWaves * i;
            if (t < P.tiles_c) {
                const v4f v = *reinterpret_cast<const v4f*>(L.v_e + 16 * t + 4 * g);
#pragma unroll
                for (int n = 0; n < 2; n++) acc[n] += v[0] * col_c[i][n][0] + v[1] * col_c[i][n][1] + v[2] * col_c[i][n][2] + v[3] * col_c[i][n][3];
            }
        }
        if (TF1 > 0 && F > 0) {
#pragma unroll
            for (int i = 0; i < TF1; i++) {
                const int t = wv + kCoopWaves * i;
                if (t < P.tiles_f) {
                    const v4f v = *reinterpret_cast<const v4f*>(L.v_e + fine_e0 + 16 * t + 4 * g);
#pragma unroll
                    for (int n = 0; n < 2; n++) acc[n] += v[0] * col_f[i][n][0] + v[1] * col_f[i][n][1] + v[2] * col_f[i][n][2] + v[3] * col_f[i][n][3];
                }
            }
        }
#pragma unroll
        for (int n = 0; n < 2; n++) {
            acc[n] += __shfl_xor(acc[n], 16);
            acc[n] += __shfl_xor(acc[n], 32);
        }
        if (g < 2) L.part[wv * 32 + 16 * g + j] = g == 0 ? acc[0] : acc[1];
        __syncthreads();
        if (tid < 32) {
            float c = L.part[tid] + L.part[32 + tid] + L.part[64 + tid];
            const float ws = L.part[kCoopWaves * 32];
            if (p.white_back) c = c + 1.f - ws;
            p.out_rgb[ray * 32 + tid] = c * 2.f - 1.f;
        }
        if (tid == 0) {
            float depth = wt_sum / w_sum;
            if (depth != depth) depth = INFINITY;
            p.out_depth[ray] = depth;
            p.out_wsum[ray] = w_sum;
        }
        __syncthreads();
    }
    if (tid == 0) range.flush(P);
}

// GNERF_MLP_AUTO: every workgroup evaluates the (cheap, deterministic) range bounds itself -- choose_mlp in render.hip -- and runs
// the body of the arithmetic they allow: one launch, no select kernel, no second grid that returns at once.
template <int TC1, int TF1, int MLP>
__global__ __launch_bounds__(kCoopThreads, 3) void render_kernel_coop(Params P) {
    extern __shared__ __align__(16) float smem[];
    if constexpr (MLP == kMlpAuto) {
        bool sp_direct;
        if (choose_mlp(P, smem, &sp_direct) == kMlpF32) render_coop_body<TC1, TF1, kMlpF32>(P, smem);
        else                                           render_coop_body<TC1, TF1, kMlpF16x3>(P, smem, sp_direct);
    } else {
        render_coop_body<TC1, TF1, MLP>(P, smem);
    }
}
