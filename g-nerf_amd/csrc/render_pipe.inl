// Pipelined render kernel: 3 SHADER WAVES + 1 SCALAR WAVE per workgroup, three rays in flight.
// Included by render.hip (inside its anonymous namespace) after render_coop.inl.
//
// In the cooperative kernel every ray alternates between phases that use all three waves (tri-plane lookups +
// MLP) and phases only one wave can do (ray march, cdf, inverse-cdf, merge), with a workgroup barrier between
// them: ablation (tools/ablate.py) shows ~35 % of its time is the single-wave phases and their bubbles.
// Here the per-sample scalar work of a ray runs on a dedicated fourth wave WHILE the three shader waves
// shade another ray, so shader waves never wait for it:
//
//   step 2k+2:  shaders  A(k+1) coarse lookups+MLP        | scalar  D(k-1) merge + final march,  out(k-2)
//   step 2k+3:  shaders  acc(k-1) colours, C(k) fine pass  | scalar  B(k+1) coarse march + importance,  P(k+2) depth proposals
//
// with ONE workgroup barrier per step (the hand-offs inside a tile are wave-private, see coop_shade_tile<false>).
// Rays r-1, r, r+1, r+2 are live at once: their per-sample scalars sit in a ring of four LDS slots, their
// colours in shader-wave registers (3 coarse sets + 1 fine set = 32 VGPRs).  The scalar wave issues its global
// loads (noise, ray) at the start of a step and consumes them at the end, behind the march/cdf work.
// Workgroups are persistent-style: each takes a contiguous run of the locality-ordered ray sequence, so the
// 3-step pipeline fill/drain is paid once per ~85 rays, not once per 16.
// Instantiated for one 16-sample tile per shader wave and pass (depth_resolution <= 48, 1 <= importance <= 48: the
// reference's training/inference default) and for two (up to 96+96: gen_videos.py's doubled counts, the ShapeNet config's
// 64+64); other shapes use the coop / generic kernels.

#ifdef GNERF_STAMPS
#define GNERF_DBG_PTR(x) ((float*)nullptr)
#else
#define GNERF_DBG_PTR(x) (x)
#endif

#ifndef GNERF_PIPE_UNIT
#define GNERF_PIPE_UNIT 8
#endif
constexpr int kPipeUnit = GNERF_PIPE_UNIT;      // rays dealt to a workgroup at a time (see render_kernel_pipe)
constexpr int kPipeThreads = 256;
constexpr int kPipeSlots = 4;
// TP = 16-sample tiles per shader wave and pass: 1 covers up to 48+48 samples (the reference's default), 2 up to 96+96, 3 up to 144+144
// (gen_videos.py doubles the ShapeNet configuration's 64+64 to 128+128)
// (gen_videos.py doubles the counts, gen_videos.py:127-128; the ShapeNet config uses 64+64).
template <int TP> struct PipeDims {
    static constexpr int kMaxS = 48 * TP;                  // samples per pass
    static constexpr int kSPad = 2 * kMaxS;                // coarse [0, kMaxS) + fine [kMaxS, 2 kMaxS)
    static constexpr int kRounds = (kMaxS + 63) / 64;      // lanes x rounds cover the samples of a pass
    static constexpr int kSlotFloats = 7 * kSPad + 2 * kMaxS + 96 + 16;        // seven per-sample arrays, cdf + fine noise (per pass), partials, ray
};

struct PipeSlot {
    float* t_e; float* sig_e; float* v_e; int* rank_e; float* s_t; float* s_sig; float* w_s; float* cdf;
    float* nf;      // [kMaxS] fine noise of this ray
    float* part;    // [3][32] colour partial sums of the shader waves
    float* misc;    // [0..11] the ray per plane: (ou, du, ov, dv) x 3 (CoopRay, render_coop.inl)  [12] item (int)  [13] ray (int)  [14] w_sum  [15] wt_sum
};
constexpr int kMiscItem = 12, kMiscRay = 13, kMiscWsum = 14, kMiscWtsum = 15;
// lane l < 12 of the scalar wave carries component misc_comp(l) of (origin xyz, direction xyz): plane l / 4, then ou, du, ov, dv
__device__ __forceinline__ int misc_comp(int l) {
    const int pl = l >> 2, q = l & 3;
    const int axis = q < 2 ? (pl == 2 ? 2 : 0) : (pl == 0 ? 1 : (pl == 1 ? 2 : 0));       // u: x, x, z   v: y, z, x
    return (q & 1) * 3 + axis;
}

template <int TP>
__device__ __forceinline__ PipeSlot pipe_slot(float* base, int slot) {
    typedef PipeDims<TP> D;
    float* p = base + slot * D::kSlotFloats;
    PipeSlot s;
    s.t_e = p; s.sig_e = p + D::kSPad; s.v_e = p + 2 * D::kSPad; s.rank_e = reinterpret_cast<int*>(p + 3 * D::kSPad);
    s.s_t = p + 4 * D::kSPad; s.s_sig = p + 5 * D::kSPad; s.w_s = p + 6 * D::kSPad; s.cdf = p + 7 * D::kSPad;
    s.nf = s.cdf + D::kMaxS; s.part = s.nf + D::kMaxS; s.misc = s.part + 96;       // cdf: kMaxS entries (n_w + 1 <= S - 2 used, the rest +inf / histogram)
    return s;
}

__host__ __device__ inline size_t pipe_lds_floats(int tp, int mlp, bool gen = false) {
    const size_t slot_floats = tp == 1 ? PipeDims<1>::kSlotFloats : (tp == 2 ? PipeDims<2>::kSlotFloats : PipeDims<3>::kSlotFloats);
    return size_t(weight_floats(mlp)) + 64 + 36 + size_t(kPipeSlots) * slot_floats + 3 * 16 * kStagePitch       // (tap records live in the staging rows)
           + kPipeUnit * 8                                                                                         // GEN: the dealing unit's rays
           + 4;                                                                                                    // GNERF_PIPE_FLAGS: the four progress counters
    (void)gen;
}

// position `seq` of the locality-ordered ray sequence -> ray index (or -1 past the end)
// PADDED: instantiations of the backward only (linear_pad(P) is never set on a forward call: the forward kernels do not carry the branch)
template <bool PADDED = false>
__device__ __forceinline__ int pipe_seq_to_ray(const Params& P, int64_t seq) {
    const gnerf_render_params& p = P.p;
    if (P.tiles_per_item > 0) {
        const int tile = int(seq >> 4), rr = int(seq & 15);
        const int item = tile / P.tiles_per_item, tt = tile % P.tiles_per_item;
        const int tx = tt / P.tiles_y, ty = tt % P.tiles_y;
        return item * p.rays_per_item + (ty * 4 + (rr >> 2)) * p.image_width + tx * 4 + (rr & 3);
    }
    if constexpr (PADDED) {
        if (const int pad = linear_pad(P); pad > 0) {         // (the staged backward of ragged calls: items padded to whole 16-ray tiles)
            const int item = int(unsigned(seq) / unsigned(pad)), local = int(unsigned(seq) - unsigned(item) * unsigned(pad));     // (seq < 2^31: checked by the launcher)
            return (item < p.n_items && local < p.rays_per_item) ? item * p.rays_per_item + local : -1;
        }
    }
    return seq < P.total_rays ? int(seq) : -1;
}

#ifndef GNERF_SCALAR_PRIO
#define GNERF_SCALAR_PRIO 3
#endif
#ifndef GNERF_PIPE2_WAVES_PER_SIMD
#define GNERF_PIPE2_WAVES_PER_SIMD 3
#endif
#ifndef GNERF_PIPE_WAVES_PER_SIMD
#define GNERF_PIPE_WAVES_PER_SIMD 4
#endif
#ifndef GNERF_PIPE_ROTATE
#define GNERF_PIPE_ROTATE 1
#endif
// GNERF_PIPE_FLAGS (round 6 experiment): the step barriers as role-to-role hand-offs through four LDS counters instead of s_barrier.
// A half-step of a shader wave depends on the SCALAR wave's previous half-step only (depth proposals, importance depths, colour weights),
// and the scalar wave's on all three shader waves' previous one (densities, colour partials); the three shader waves never exchange
// anything.  s_barrier makes each of them wait for the slowest of the three every half-step (stamps: 11 % of a shader wave's time);
// with counters a shader wave that is done goes on as soon as the scalar wave has finished its part -- which it has, it is the one
// that waits (38 % of its time) -- so the three SIMDs' different loads average out over two half-steps instead of none.
#ifndef GNERF_PIPE_FLAGS
#define GNERF_PIPE_FLAGS 0
#endif
// FULL: the call fills the kernel's sample slots exactly (depth_resolution = depth_resolution_importance = 48 TP: the reference's 48+48
// default, gen_videos.py's doubled 96+96) with plain stratified sampling (no disparity spacing, no per-ray limits) and no stage dump.
// Sample counts, tile counts and every `k < S` predicate are then compile-time constants; the cold options are not compiled in at
// all.  Same arithmetic, same results (the tests run both instantiations on the same inputs); what it buys is the scalar wave's
// instruction count and the scalar-register pressure: the general instantiation spills 227 SGPRs to VGPR lanes and reloads them
// with v_readlane_b32 all over the scalar wave's loop.
// GEN: the instantiation that can make its rays and its uniform draws itself (gnerf_render_params.cam2world / rng_mode, ABI 8), each
// chosen per launch: rays from the item's camera with gnerf_make_rays' arithmetic, draws as torch's device generator would have
// made them (raygen.h).  Both run on the scalar wave, whose SIMD is the half-idle one: a ray's 96 draws are two
// wave-wide Philox evaluations (~100 integer instructions each) in place of two loads, its direction is computed for a whole dealing
// unit at a time (one lane per ray of the unit, parked in 256 bytes of LDS).
// BWD: the first pass of the renderer's backward (gnerf_render_backward, staged form) -- the same forward pipeline, but instead of
// compositing colours it hands the backward's second kernel (render_bwd_tiles_kernel, render_bwd.inl) what it needs per sample, in
// the ray's merged depth order: the depth, the colour weight v_r and dL/dsigma_r.  Shader waves reduce each sample's colours to
// q = sum_c dL/dcolour[c] colour[c] (all the composite's gradient needs of them); the scalar wave runs the composite's gradient
// (ray_marcher.py:25-57 backwards) as wave scans right after the merge.  Nothing else of the ray is kept.
template <int TP, int MLP, bool FULL, bool GEN, bool BWD = false>
__device__ __forceinline__ void render_pipe_body(const Params& P, float* smem, const gnerf_render_grads* Gr = nullptr, float* bstage = nullptr, bool sp_direct = false) {
    typedef PipeDims<TP> D;
    constexpr int kPipeMaxS = D::kMaxS, kPipeSPad = D::kSPad, kSlotFloats = D::kSlotFloats, RND = D::kRounds;
    const gnerf_render_params& p = P.p;
    const int tid = threadIdx.x, lane = tid & 63;
#if defined(GNERF_ROLE_ROT)
    // experiment: which SIMD hosts the scalar wave.  Physical wave w of a workgroup sits on SIMD w; role = (w + rot) & 3 puts the scalar
    // role of CU-mates on different SIMDs when their `rot` differ.  1: rot from the launch position (blockIdx / 256), 2: blockIdx / 8,
    // 3: from the wave slot the hardware gave wave 0 (HW_ID.WAVE_ID)
    int rot;
    if (GNERF_ROLE_ROT == 1) rot = (blockIdx.x >> 8) & 3;
    else if (GNERF_ROLE_ROT == 2) rot = (blockIdx.x >> 3) & 3;
    else {
        if (tid == 0) { unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); reinterpret_cast<unsigned*>(smem)[0] = hw & 15u; }
        __syncthreads();
        rot = reinterpret_cast<const unsigned*>(smem)[0] & 3;
        __syncthreads();
    }
    const int wv = __builtin_amdgcn_readfirstlane(((tid >> 6) + rot) & 3);
#else
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
    const int S = FULL ? kPipeMaxS : p.depth_resolution, F = FULL ? kPipeMaxS : p.depth_resolution_importance;
    const int tiles_c = FULL ? 3 * TP : P.tiles_c, tiles_f = FULL ? 3 * TP : P.tiles_f;
    float* const debug = (FULL || BWD) ? nullptr : GNERF_DBG_PTR(p.debug);       // (the backward's params carry no debug buffer)
    // fixed slot layout: coarse samples at [0,48), fine samples at [48,96); unused entries hold +inf depths so that the
    // fully unrolled 4-wide key scans below need no bounds checks
    constexpr int fine_e0 = kPipeMaxS, s_pad = kPipeSPad;
    const int n_all = S + F;
    CoopLds L;
    float* slots = smem + weight_floats(MLP) + 64 + 36;
    // A wave's tap records (16 records of kFwdTapStride = 28 dwords, 24 of them used) and its staging rows (16 rows of kFwdStagePitch
    // = 32 dwords, swizzled) share one area of 16 x kStagePitch dwords.  What makes that safe is an ORDER inside coop_shade_tile, not a
    // layout: every read of a record -- the offsets of both lookup steps and, last, step 1's weights (`w1`, fetched right before `row0`
    // is stored and kept in front of that store by pin()) -- is issued before the first row is written, and the rows have been read back
    // (f_lo / f_hi) before the next tile's records are written; LDS operations of one wave execute in issue order.  Both layouts must fit
    // the area whatever tools/build_variants.sh sets GNERF_TAP_STRIDE / GNERF_STAGE_SWZ to.  Those 4.6 KB are what lets FOUR workgroups
    // share a CU's 160 KB.
    static_assert(16 * kFwdTapStride <= 16 * kStagePitch && 16 * kFwdStagePitch <= 16 * kStagePitch && kFwdTapStride >= kTapDwords,
                  "tap records and staging rows share 16 x kStagePitch dwords per wave");
    L.taps = slots + kPipeSlots * kSlotFloats;
    L.stage = L.taps;
    L.wave_pitch_taps = L.wave_pitch_stage = 16 * kStagePitch;

    // This workgroup's share of the locality-ordered ray sequence.  Workgroups b, b+8, ... share an XCD (round-robin
    // dispatch), and each XCD owns a contiguous eighth of the sequence.  Inside an XCD the sequence is dealt to its W
    // workgroups in UNITS of kPipeUnit consecutive rays, round-robin: at any moment the W workgroups are within a few
    // units of each other, i.e. on neighbouring rays, so the XCD's 4 MB L2 holds the texels they share.  (Giving each
    // workgroup one long contiguous run instead spreads the XCD over 96 distant image regions: measured 1.3 GB of L2
    // misses per launch; dealing units of 8 rays is 10 % faster, 4-16 are within 3 % of each other.)  Speed only -- any assignment is correct.
    const int64_t total_seq = (P.tiles_per_item > 0 || (BWD && linear_pad(P) > 0)) ? int64_t(P.n_tiles) * 16 : int64_t(P.total_rays);
    const int W = gridDim.x / kNumXCD, xcd = blockIdx.x % kNumXCD, wg = blockIdx.x / kNumXCD;
    const int64_t x0 = total_seq * xcd / kNumXCD, x1 = total_seq * (xcd + 1) / kNumXCD;
    const int unit = P.pipe_unit;
    const int n_units = int((x1 - x0 + unit - 1) / unit);
    const int my_units = n_units > wg ? (n_units - wg + W - 1) / W : 0;
    const int nr = my_units * unit;
    auto local_to_ray = [&](int r) -> int {
        const int64_t seq = x0 + (int64_t(wg) + int64_t(r / unit) * W) * unit + r % unit;
        return seq < x1 ? pipe_seq_to_ray<BWD>(P, seq) : -1;
    };

    stage_decoder<MLP>(L, smem, p, tid, kPipeThreads);

    Stamps st;
    // ------------------------------------------------------------------ scalar-wave pieces (lambdas, wave 3 only)
    DepthRange range;
    float pre_uc[RND] = {}, pre_uf[RND] = {}, pre_ray = 0.f, pre_rs = 0.f, pre_re = 0.f;       // prefetched by propose_issue
    float pre_g = 0.f;                                                                         // BWD: dL/d(colour sum) [0..31], dL/ddepth [32], dL/dwsum [33]
    int pre_ray_id = -1;

    float* const unit_rays = L.taps + 3 * 16 * kStagePitch;        // GEN: [kPipeUnit][8] origin, direction of the current dealing unit's rays
    auto propose_issue = [&](int r) {           // P(r), first half: start the global loads (or make the values: GEN)
        pre_ray_id = (r >= 0 && r < nr) ? local_to_ray(r) : -1;
        if (GEN && p.cam2world && r >= 0 && r < nr && r % unit == 0 && lane < unit) {
            // rays of this dealing unit, one lane each (r .. r + unit - 1 are consecutive positions of the sequence)
            const int64_t seq = x0 + (int64_t(wg) + int64_t(r / unit) * W) * unit + lane;
            const int id = seq < x1 ? pipe_seq_to_ray<BWD>(P, seq) : -1;
            if (id >= 0) {
                const int item = id / p.rays_per_item, m = id - item * p.rays_per_item;
                const float* M = p.cam2world + item * 16;
                float d[3];
                camera_ray(M, p.intrinsics + item * 9, p.image_width, m / p.image_width, m % p.image_width, d);
#pragma unroll
                for (int c = 0; c < 3; c++) { unit_rays[lane * 8 + c] = M[c * 4 + 3]; unit_rays[lane * 8 + 3 + c] = d[c]; }
            }
        }
        if (pre_ray_id < 0) return;
        const int64_t ray = pre_ray_id;
        if (GEN && p.rng_mode) {
            TorchRandDraw dc = P.draw_c, df = P.draw_f;
            uint64_t first_c = uint64_t(ray) * S, first_f = uint64_t(ray) * F;
            if (p.rng_per_item) {                                   // the draws of a call of this item alone
                const int item = pre_ray_id / p.rays_per_item;
                const uint64_t in_item = uint64_t(pre_ray_id - item * p.rays_per_item);
                first_c = in_item * S; first_f = in_item * F;
                dc.ctr += uint64_t(item) * P.draw_item_ctr; df.ctr += uint64_t(item) * P.draw_item_ctr;
            }
#pragma unroll
            for (int q = 0; q < RND; q++) {
                if (lane + 64 * q < S) pre_uc[q] = torch_rand_element(dc, first_c + lane + 64 * q);
                if (lane + 64 * q < F) pre_uf[q] = torch_rand_element(df, first_f + lane + 64 * q);
            }
        } else {
#pragma unroll
            for (int q = 0; q < RND; q++) {
                if (lane + 64 * q < S) pre_uc[q] = p.noise_coarse[ray * S + lane + 64 * q];
                if (lane + 64 * q < F) pre_uf[q] = p.noise_fine[ray * F + lane + 64 * q];
            }
        }
        if (GEN && p.cam2world) {
            if (lane < 12) pre_ray = unit_rays[(r % unit) * 8 + misc_comp(lane)];       // (LDS operations of one wave execute in order)
        } else {
            if (lane < 12) { const int c = misc_comp(lane); pre_ray = c < 3 ? p.ray_origins[ray * 3 + c] : p.ray_dirs[ray * 3 + c - 3]; }
        }
        if (!FULL && p.ray_start_per_ray) { pre_rs = p.ray_start_per_ray[ray]; pre_re = p.ray_end_per_ray[ray]; }
        if constexpr (BWD) {        // the ray's incoming gradients: rgb = 2 * composite - 1 (ray_marcher.py:55), depth, weight sum
            pre_g = 0.f;
            if (lane < 32) { if (Gr->grad_rgb) pre_g = 2.f * Gr->grad_rgb[ray * 32 + lane]; }
            else if (lane == 32) { if (Gr->grad_depth) pre_g = Gr->grad_depth[ray]; }
            else if (lane == 33) { if (Gr->grad_wsum) pre_g = Gr->grad_wsum[ray]; }
        }
    };
    auto propose_finish = [&](int r) {          // P(r), second half: depth proposals (renderer.py:169-192) into the slot
        if (r < 0 || r >= nr) return;
        PipeSlot sl = pipe_slot<TP>(slots, r & (kPipeSlots - 1));
        if (lane == kMiscRay) sl.misc[kMiscRay] = __int_as_float(pre_ray_id);
        if (pre_ray_id < 0) return;
#pragma unroll
        for (int q = 0; q < RND; q++) {
        const int k = lane + 64 * q;
        if (k < S) {
            const float u = pre_uc[q];
            float d;
            if (!FULL && p.disparity_space_sampling) {
                const float step = 1.0f / float(S - 1);
                const float lin = (k < S / 2) ? __fmul_rn(step, float(k)) : __fsub_rn(1.0f, __fmul_rn(step, float(S - 1 - k)));
                const float q = __fadd_rn(lin, __fmul_rn(u, P.disp_delta));
                d = __fdiv_rn(1.0f, __fadd_rn(__fmul_rn(P.inv_start, __fsub_rn(1.0f, q)), __fmul_rn(P.inv_end, q)));
            } else if (!FULL && p.ray_start_per_ray) {
                const float span = __fsub_rn(pre_re, pre_rs);
                const float lin = __fadd_rn(pre_rs, __fmul_rn(__fdiv_rn(float(k), float(S - 1)), span));
                d = __fadd_rn(lin, __fmul_rn(u, __fdiv_rn(span, float(S - 1))));
            } else {
                const float step = __fdiv_rn(__fsub_rn(p.ray_end, p.ray_start), float(S - 1));
                const float lin = (k < S / 2) ? __fadd_rn(p.ray_start, __fmul_rn(step, float(k)))
                                              : __fsub_rn(p.ray_end, __fmul_rn(step, float(S - 1 - k)));
                d = __fadd_rn(lin, __fmul_rn(u, P.delta));
            }
            sl.t_e[k] = d;
            if (debug) debug[(int64_t(pre_ray_id) * GNERF_DEBUG_SLOTS + GNERF_DBG_DEPTH_COARSE) * n_all + k] = d;
        }
        }
#pragma unroll
        for (int q = 0; q < RND; q++)
            if (lane + 64 * q < F) sl.nf[lane + 64 * q] = pre_uf[q];
        if (lane < 12) sl.misc[lane] = pre_ray;
        if (lane == kMiscItem) sl.misc[kMiscItem] = __int_as_float(pre_ray_id / p.rays_per_item);
        if constexpr (BWD) {
            if (lane < 34) sl.part[lane] = pre_g;            // [0..31] dL/d(colour sum), [32] dL/ddepth, [33] dL/dwsum (the forward's other partials are unused here)
        }
        for (int e = lane; e < s_pad; e += 64) {
            sl.v_e[e] = 0.f;
            if ((e >= S && e < fine_e0) || e >= fine_e0 + F) sl.t_e[e] = INFINITY;      // tile padding sorts last
        }
    };
    auto importance = [&](int r) {              // B(r): coarse march (ray_marcher.py:26-42) + importance depths (renderer.py:194-253)
        if (r < 0 || r >= nr) return;
        PipeSlot sl = pipe_slot<TP>(slots, r & (kPipeSlots - 1));
        const int ray_id = __float_as_int(sl.misc[kMiscRay]);
        if (ray_id < 0) return;
        float* dbg = debug ? debug + int64_t(ray_id) * GNERF_DEBUG_SLOTS * n_all : nullptr;
#ifdef GNERF_ABLATE_PIPESCALAR  // timing-only build: the scalar wave does NONE of its per-ray passes (fine depths = coarse depths; outputs are wrong):
        // the upper bound of what any reformulation of them -- the lane-per-ray pass included -- could return
        for (int i = lane; i < F; i += 64) { sl.t_e[fine_e0 + i] = sl.t_e[min(i, S - 1)]; sl.rank_e[fine_e0 + i] = 0; }
        (void)dbg;
        return;
#endif
        float ws, wts;
        march(sl.t_e, sl.sig_e, sl.w_s, S, lane, ws, wts);
        lds_wave_sync();
        const int n_w = S - 3;
        float pw[RND], carry = 0.f;
#pragma unroll
        for (int q = 0; q < RND; q++) {
            const int i = lane + 64 * q;
            pw[q] = 0.f;
            if (i < n_w) {
                const float w0 = sl.w_s[i], w1 = sl.w_s[i + 1], w2 = sl.w_s[i + 2];
                pw[q] = ((max_nn(w0, w1) + max_nn(w1, w2)) * 0.5f + 0.01f) + 1e-5f;
            }
            carry = wave_last(wave_scan_add(pw[q], lane) + carry);
        }
        const float inv_total = __builtin_amdgcn_rcpf(carry);
        // cdf[i+1] = cumsum(pw * inv_total): scan the normalised terms like the reference does (pdf first, then cumsum)
        carry = 0.f;
#pragma unroll
        for (int q = 0; q < RND; q++) {
            const int i = lane + 64 * q;
            const float incl = wave_scan_add(pw[q] * inv_total, lane) + carry;
            if (i < n_w) sl.cdf[i + 1] = incl;
            carry = wave_last(incl);
        }
        if (lane == 0) sl.cdf[0] = 0.f;
        for (int e = n_w + 1 + lane; e < kPipeMaxS; e += 64) sl.cdf[e] = INFINITY;           // lets the count below read 4 at a time
        lds_wave_sync();
#pragma unroll
        for (int q = 0; q < RND; q++) {
            const int i = lane + 64 * q;
            if (i < F) {
                const float u = sl.nf[i];
                // searchsorted(cdf, u, right=True) = #{cdf <= u}.  The cdf ascends (a running sum of positive terms; its tail is +inf),
                // so the count is found by bisection: log2 steps of one LDS word each instead of a compare + add-with-carry per key
                // (96 vector instructions and 12 wide LDS reads per ray at 48 keys).  The reads are dependent, ~64 cycles apiece, on the
                // wave that has the slack for them (tools/stamps.py: a third of its time is spent waiting at the step barriers).
                int cnt = 0;
#pragma unroll
                for (int step = (kPipeMaxS >= 64 ? (kPipeMaxS >= 128 ? 128 : 64) : 32); step > 0; step >>= 1) {
                    const int probe = cnt + step;                          // is cdf[probe - 1] <= u, i.e. are there at least `probe` such entries?
                    if (probe <= kPipeMaxS && sl.cdf[probe - 1] <= u) cnt = probe;
                }
                const int below = max(cnt - 1, 0), above = min(cnt, n_w);
                sl.rank_e[fine_e0 + i] = below;                            // the sample's bin, for the merge (finalize)
                const float cb = sl.cdf[below], ca = sl.cdf[above];
                const float bb = (sl.t_e[below] + sl.t_e[below + 1]) * 0.5f;
                const float ba = (sl.t_e[above] + sl.t_e[above + 1]) * 0.5f;
                float denom = ca - cb;
                if (denom < 1e-5f) denom = 1.f;
                const float d = bb + (u - cb) * __builtin_amdgcn_rcpf(denom) * (ba - bb);
                sl.t_e[fine_e0 + i] = d;
                if (dbg) dbg[GNERF_DBG_DEPTH_FINE * n_all + i] = d;
            }
        }
        if (dbg) {
            for (int k = lane; k < S; k += 64) dbg[GNERF_DBG_SIGMA_COARSE * n_all + k] = sl.sig_e[k];
            for (int k = lane; k < S - 1; k += 64) dbg[GNERF_DBG_WEIGHT_COARSE * n_all + k] = sl.w_s[k];
        }
        // the cdf is dead from here until this slot's next ray: it becomes the merge's per-bin histogram (finalize), zeroed now
        // (LDS operations of one wave execute in issue order, so these stores follow the reads above without a fence)
        for (int e = lane; e < kPipeMaxS; e += 64) reinterpret_cast<int*>(sl.cdf)[e] = 0;
    };
    auto finalize = [&](int r) {                // D(r): merge by depth (renderer.py:157-167) + final march + per-sample colour weights
        if (r < 0 || r >= nr) return;
        PipeSlot sl = pipe_slot<TP>(slots, r & (kPipeSlots - 1));
        const int ray_id = __float_as_int(sl.misc[kMiscRay]);
        if (ray_id < 0) return;
        float* dbg = debug ? debug + int64_t(ray_id) * GNERF_DEBUG_SLOTS * n_all : nullptr;
        // Stable rank in cat([coarse, fine]).  Coarse depths ascend by construction, so
        //   rank(coarse k) = k + #{fine < t_k}            rank(fine i) = #{coarse <= t_i} + #{fine before i}
        // (ties: coarse first, then lower index -- what a stable sort of the concatenation gives).
        // The two coarse-vs-fine counts need no scan.  A fine sample drawn from bin b of the inverse cdf lies between the
        // midpoints mid[b] and mid[b+1] of the coarse depths (renderer.py:243-252; `below` of sample_pdf, kept by importance()),
        // i.e. between coarse samples b and b+2 up to an ulp of rounding: coarse samples < b are before it and coarse samples
        // > b+2 after it for certain (three consecutive coarse depths cannot be within rounding of each other: neighbours need
        // a jitter of ~1 followed by a jitter of 0), and only b, b+1, b+2 are compared.  The coarse side is the mirror image:
        // fine samples of bins <= k-3 are before coarse sample k, bins >= k+1 after it, and for the bins k-2, k-1, k the
        // answer is the complement of the fine side's three comparisons, gathered per bin in a packed LDS histogram
        // (count | #before coarse b | #before b+1 | #before b+2, 8 bits each) and prefix-summed over the bins.
        // "#fine before i" is a compare + add-with-carry scan over the fine keys read four at a time as LDS broadcasts, first
        // taken as #{fine < t_i}; two fine samples with bit-identical depth then collide on a rank, which is detected through
        // an owner table and repaired by a tie-broken recount (rare: needs two equal uniform draws or rounding collisions).
#ifdef GNERF_ABLATE_PIPESCALAR
        if constexpr (!BWD) {
            for (int q = lane; q < n_all; q += 64) sl.v_e[q < S ? q : fine_e0 + (q - S)] = 1.f / float(n_all);
            if (lane == 0) { sl.misc[kMiscWsum] = 1.f; sl.misc[kMiscWtsum] = sl.t_e[0]; range.add(P, __float_as_int(sl.misc[kMiscItem]), sl.t_e[0], sl.t_e[S - 1]); }
            (void)dbg;
            return;
        }
#endif
        int rank_f[RND], rank_c[RND];
        float key_f[RND], key_c[RND];
        float q_c[RND] = {}, q_f[RND] = {};           // BWD: this lane's samples' q, read before v_e is reused below
        if constexpr (BWD) {
#pragma unroll
            for (int q = 0; q < RND; q++) {
                const int i = lane + 64 * q;
                if (i < S) q_c[q] = sl.v_e[i];
                if (i < F) q_f[q] = sl.v_e[fine_e0 + i];
            }
        }
#ifdef GNERF_ABLATE_PIPERANK    // timing-only build: no rank merge on the scalar wave (coarse then fine in index order; outputs are wrong)
#pragma unroll
        for (int q = 0; q < RND; q++) {
            const int i = lane + 64 * q;
            rank_f[q] = S + i; key_f[q] = i < F ? sl.t_e[fine_e0 + i] : 0.f;
            rank_c[q] = i; key_c[q] = i < S ? sl.t_e[i] : 0.f;
            if (i < F) sl.rank_e[fine_e0 + i] = S + i;
            if (i < S) sl.rank_e[i] = i;
        }
#else
#pragma unroll
        for (int q = 0; q < RND; q++) {
            const int i = lane + 64 * q;
            rank_f[q] = 0; key_f[q] = 0.f;
            if (i < F) {
                const float key = sl.t_e[fine_e0 + i];
                int rk = 0;
#pragma unroll 4
                for (int o2 = 0; o2 < kPipeMaxS; o2 += 4) {
                    const v4f k4 = *reinterpret_cast<const v4f*>(sl.t_e + fine_e0 + o2);
#pragma unroll
                    for (int c2 = 0; c2 < 4; c2++) rk += (k4[c2] < key) ? 1 : 0;
                }
                {
                    const int b = sl.rank_e[fine_e0 + i];
                    const int c0 = sl.t_e[b] <= key ? 1 : 0, c1 = sl.t_e[b + 1] <= key ? 1 : 0, c2 = sl.t_e[b + 2] <= key ? 1 : 0;
                    rk += b + c0 + c1 + c2;
                    atomicAdd(reinterpret_cast<int*>(sl.cdf) + b, 1 | ((1 - c0) << 8) | ((1 - c1) << 16) | ((1 - c2) << 24));
                }
                sl.rank_e[fine_e0 + i] = rk;
                reinterpret_cast<int*>(sl.w_s)[rk] = i;             // owner table (w_s is free until the final march)
                rank_f[q] = rk; key_f[q] = key;
            }
        }
        lds_wave_sync();                        // the histogram is complete
        {
            const int* hist = reinterpret_cast<const int*>(sl.cdf);
            float carry = 0.f;
#pragma unroll
            for (int q = 0; q < RND; q++) {     // #fine in bins <= k, parked in v_e[k] (free until the end of this function, and rewritten there)
                const int k = lane + 64 * q;
                const float incl = wave_scan_add(float(hist[min(k, kPipeMaxS - 1)] & 255), lane) + carry;
                carry = wave_last(incl);
                if (k < S) sl.v_e[k] = incl;
            }
            lds_wave_sync();
#pragma unroll
            for (int q = 0; q < RND; q++) {
                const int k = lane + 64 * q;
                rank_c[q] = k; key_c[q] = 0.f;
                if (k < S) {
                    const float key = sl.t_e[k];
                    int rk = k;
                    // "ascend by construction" holds up to rounding: t_k = lin_k + u delta can round one ulp past t_{k+1} when
                    // u is within 1e-5 of 1.  Only neighbours can swap (the grid step is ~1e5 ulps), so the number of coarse
                    // samples sorted before k is k +- 1 from two compares.
                    if (k + 1 < S && sl.t_e[k + 1] < key) rk += 1;
                    if (k > 0 && sl.t_e[k - 1] > key) rk -= 1;
                    const int h0 = hist[k], h1 = hist[max(k - 1, 0)], h2 = hist[max(k - 2, 0)];       // bins >= S-2 do not exist: zero
                    rk += (k >= 3 ? int(sl.v_e[k - 3]) : 0) + ((h0 >> 8) & 255) + (k >= 1 ? (h1 >> 16) & 255 : 0) + (k >= 2 ? (h2 >> 24) & 255 : 0);
                    sl.rank_e[k] = rk;
                    rank_c[q] = rk; key_c[q] = key;
                }
            }
        }
        lds_wave_sync();
        bool clash = false;
#pragma unroll
        for (int q = 0; q < RND; q++) clash = clash || (lane + 64 * q < F && reinterpret_cast<const int*>(sl.w_s)[rank_f[q]] != lane + 64 * q);
        if (__any(clash)) {                                         // wave-uniform, rare
#pragma unroll
            for (int q = 0; q < RND; q++) {
                const int i = lane + 64 * q;
                if (i < F) {
                    int fix = 0;
#ifndef GNERF_CLASH_UNROLL      // (rare path: unrolled at the compile-time counts it hoists 48 lane masks out of the ray loop -- 96 spilled SGPRs)
#pragma nounroll
#endif
                    for (int o2 = 0; o2 < F; o2++) fix += (sl.t_e[fine_e0 + o2] == key_f[q] && o2 < i) ? 1 : 0;
                    rank_f[q] += fix;
                    sl.rank_e[fine_e0 + i] = rank_f[q];
                }
            }
        }
#endif
#pragma unroll
        for (int q = 0; q < RND; q++) {
            const int i = lane + 64 * q;
            if (i < F) { sl.s_t[rank_f[q]] = key_f[q]; sl.s_sig[rank_f[q]] = sl.sig_e[fine_e0 + i]; }
            if (i < S) { sl.s_t[rank_c[q]] = key_c[q]; sl.s_sig[rank_c[q]] = sl.sig_e[i]; }
            if constexpr (BWD) {                // q in depth order; the histogram (cdf) and the fine noise behind it are dead: 2 kMaxS floats
                if (i < F) sl.cdf[rank_f[q]] = q_f[q];
                if (i < S) sl.cdf[rank_c[q]] = q_c[q];
            }
        }
        lds_wave_sync();
        GNERF_STAMP(st, 13);    // merge ranks
        if constexpr (BWD) {
            // ---- final march keeping the transmittance in front of every interval (ray_marcher.py:26-42), then the composite's gradient
            // (the arithmetic of render_bwd_kernel, render_bwd.inl).  Per-interval arrays go to this slot's dead element arrays:
            // trans -> t_e, dL/d(interval density input) -> sig_e; the weights to w_s as in the forward.
            const int n_int = n_all - 1;
            float* const s_q = sl.cdf;
            float* const trans = sl.t_e;
            float* const ds = sl.sig_e;
            float w_sum, wt_sum;
            {
                float carry = 1.f, acc_w = 0.f, acc_wt = 0.f;
                for (int kb = 0; kb < n_int; kb += 64) {
                    const int k = kb + lane;
                    const bool ok = k < n_int;
                    float alpha = 0.f, tmid = 0.f;
                    if (ok) {
                        const float t0 = sl.s_t[k], t1 = sl.s_t[k + 1];
                        const float smid = softplus_march((sl.s_sig[k] + sl.s_sig[k + 1]) * 0.5f - 1.f);
                        tmid = (t0 + t1) * 0.5f;
                        alpha = 1.f - exp_hw(-(smid * (t1 - t0)));
                    }
                    const float x = ok ? (1.f - alpha + 1e-10f) : 1.f;
                    const float incl = wave_scan_mul(x, lane);
                    const float tr = wave_shift_up(incl, 1.f) * carry;
                    carry *= wave_last(incl);
                    if (ok) { const float wk = alpha * tr; sl.w_s[k] = wk; trans[k] = tr; acc_w += wk; acc_wt += wk * tmid; }
                }
                w_sum = wave_sum(acc_w);
                wt_sum = wave_sum(acc_wt);
            }
            lds_wave_sync();
            //   dL/dw_k = (q_k + q_{k+1}) / 2 - [white_back] sum_c G[c] + g_depth (tmid_k - depth) / W + g_wsum
            //   dL/dalpha_k = dL/dw_k T_k - (sum_{m>k} dL/dw_m w_m) / (1 - alpha_k + 1e-10)
            const float g_depth = sl.part[32], g_wsum = sl.part[33];
            const float g_sum = p.white_back ? wave_sum(lane < 32 ? sl.part[lane] : 0.f) : 0.f;
            const float depth = wt_sum / w_sum;
            const float gd_scale = (w_sum > 0.f && depth == depth) ? g_depth / w_sum : 0.f;     // nan_to_num'd rays pass no depth gradient
            const float gw_const = g_wsum - g_sum;
            float carry = 0.f;
            for (int kb = 0; kb < n_int; kb += 64) {               // from the far end: suffix sums are prefix sums here
                const int k = n_int - 1 - (kb + lane);
                const bool ok = k >= 0;
                float gw = 0.f, wk = 0.f, t0 = 0.f, t1 = 0.f, smid_in = 0.f;
                if (ok) {
                    t0 = sl.s_t[k]; t1 = sl.s_t[k + 1];
                    smid_in = (sl.s_sig[k] + sl.s_sig[k + 1]) * 0.5f - 1.f;
                    wk = sl.w_s[k];
                    gw = (s_q[k] + s_q[k + 1]) * 0.5f + gw_const + gd_scale * ((t0 + t1) * 0.5f - depth);
                }
                const float incl = wave_scan_add(gw * wk, lane) + carry;
                carry = wave_last(incl);
                if (ok) {
                    const float after = incl - gw * wk;                      // sum over m > k
                    const float delta = t1 - t0;
                    const float dens = softplus_march(smid_in);
                    const float one_minus_alpha = exp_hw(-(dens * delta));
                    const float alpha = 1.f - one_minus_alpha;
                    const float d_alpha = gw * trans[k] - after / (1.f - alpha + 1e-10f);
                    const float d_dens = d_alpha * delta * one_minus_alpha;
                    const float e = exp_hw(-smid_in);
                    ds[k] = smid_in > 20.f ? d_dens : d_dens * __builtin_amdgcn_rcpf(1.f + e);      // softplus' = sigmoid
                }
            }
            lds_wave_sync();
            // ---- per sample, in depth order: depth, colour weight, dL/dsigma -> the staging buffer.  Depths at the head of the ray's
            // block (where plane_scatter_kernel reads them); v and dL/dsigma of the 16 ranks of tile T in the first 32 floats of the
            // tile's own dX rows, which render_bwd_tiles_kernel reads before it overwrites them with dX.
            float* const ray_block = bstage + int64_t(ray_id) * P.bwd_ray_stride;
            for (int r = lane; r < n_all; r += 64) {
                const float wl = r > 0 ? sl.w_s[r - 1] : 0.f, wr = r < n_int ? sl.w_s[r] : 0.f;
                const float dl = r > 0 ? ds[r - 1] : 0.f, dr = r < n_int ? ds[r] : 0.f;
                ray_block[r] = sl.s_t[r];
                float* tile_rows = ray_block + n_all + (r >> 4) * P.bwd_tile_pitch;
                tile_rows[r & 15] = (wl + wr) * 0.5f;            // colour of sample r enters intervals r-1 and r with weight 1/2 each
                tile_rows[16 + (r & 15)] = (dl + dr) * 0.5f;     // so does its density
            }
            return;
        }
        float ws, wts;
        march(sl.s_t, sl.s_sig, sl.w_s, n_all, lane, ws, wts);
        lds_wave_sync();
        GNERF_STAMP(st, 14);    // final march
        for (int q = lane; q < n_all; q += 64) {
            const int e = q < S ? q : fine_e0 + (q - S);
            const int rk = sl.rank_e[e];
            const float wl = rk > 0 ? sl.w_s[rk - 1] : 0.f, wr = rk < n_all - 1 ? sl.w_s[rk] : 0.f;
            sl.v_e[e] = (wl + wr) * 0.5f;                       // midpoint colours (ray_marcher.py:27) regrouped per sample
        }
        if (lane == 0) {
            sl.misc[kMiscWsum] = ws;
            sl.misc[kMiscWtsum] = wts;
            range.add(P, __float_as_int(sl.misc[kMiscItem]), sl.s_t[0], sl.s_t[n_all - 1]);
        }
        if (dbg) {
            for (int k = lane; k < F; k += 64) dbg[GNERF_DBG_SIGMA_FINE * n_all + k] = sl.sig_e[fine_e0 + k];
            for (int k = lane; k < n_all; k += 64) { dbg[GNERF_DBG_DEPTH_SORTED * n_all + k] = sl.s_t[k]; dbg[GNERF_DBG_SIGMA_SORTED * n_all + k] = sl.s_sig[k]; }
            for (int k = lane; k < n_all - 1; k += 64) dbg[GNERF_DBG_WEIGHT_FINAL * n_all + k] = sl.w_s[k];
        }
    };
    auto output = [&](int r) {                  // out(r): sum the shader waves' colour partials, write the three outputs
        if (r < 0 || r >= nr) return;
        PipeSlot sl = pipe_slot<TP>(slots, r & (kPipeSlots - 1));
        const int ray_id = __float_as_int(sl.misc[kMiscRay]);
        if (ray_id < 0) return;
        const float ws = sl.misc[kMiscWsum], wts = sl.misc[kMiscWtsum];
        if (lane < 32) {
            float c = sl.part[lane] + sl.part[32 + lane] + sl.part[64 + lane];
            if (p.white_back) c = c + 1.f - ws;
            p.out_rgb[int64_t(ray_id) * 32 + lane] = c * 2.f - 1.f;
        }
        if (lane == 0) {
            float depth = wts / ws;
            if (depth != depth) depth = INFINITY;               // nan_to_num(nan=inf); the call-wide clamp is applied by clamp_depth_kernel
            p.out_depth[ray_id] = depth;
            p.out_wsum[ray_id] = ws;
        }
    };

    // ------------------------------------------------------------------ shader-wave pieces (waves 0..2)
    // Returns whether ray r exists (a run's last dealing unit can be short).  Everything the head of the tile needs from the slot --
    // origin, direction, item, ray id and this lane's depth -- is fetched in ONE LDS round trip (two broadcast ds_read_b128 and the
    // depth word issued together); round 4 read the ray id, the item and the rest one after the other, each behind a wait and a
    // v_readfirstlane: three dependent round trips in front of every tile (tools/stamps.py "slot params": 5 % of a shader wave's time).
    auto shade = [&](int r, bool fine, v4f (&col)[TP][2]) -> bool {
        if (r < 0 || r >= nr) return false;
        PipeSlot sl = pipe_slot<TP>(slots, r & (kPipeSlots - 1));
        const float* t_list = fine ? sl.t_e + fine_e0 : sl.t_e;
        const int count = fine ? F : S;
        typedef float v2f_t __attribute__((ext_vector_type(2)));
        v4f uv = *reinterpret_cast<const v4f*>(sl.misc + 4 * min(lane >> 4, 2));              // this lane's plane: (ou, du, ov, dv)
        v2f_t ids = *reinterpret_cast<const v2f_t*>(sl.misc + kMiscItem);
        float depth0 = t_list[min(16 * wv + (lane & 15), count - 1)];                // tile wv (the first of this wave's tiles)
        // (all three reads leave together, in front of the branch on the ray id: left alone the compiler sinks the ray's words behind that
        //  branch and the item's word behind the next -- three dependent LDS round trips at the head of every tile.  The operands are the
        //  loaded register tuples themselves: as seven scalars the pin cost three v_mov_b32 per tile)
        asm volatile("" : "+v"(uv), "+v"(ids), "+v"(depth0));
        const float id_item = ids[0], id_ray = ids[1], r_ou = uv[0], r_du = uv[1], r_ov = uv[2], r_dv = uv[3];
        const int ray_id = __builtin_amdgcn_readfirstlane(__float_as_int(id_ray));
        if (ray_id < 0) return false;
        CoopRay R;
        const int item = __builtin_amdgcn_readfirstlane(__float_as_int(id_item));
        R.planes_item = reinterpret_cast<const char*>(p.planes_nhwc) + int64_t(item) * P.item_bytes;
        R.ou = r_ou; R.du = r_du; R.ov = r_ov; R.dv = r_dv;
        if constexpr (!FULL) {                                     // density noise: a cold option, like disparity sampling and per-ray limits
            const float* const sn = fine ? p.sigma_noise_fine : p.sigma_noise_coarse;
            R.sig_noise = sn ? sn + int64_t(ray_id) * count : nullptr;
        }
        GNERF_STAMP(st, 0);     // ray parameters from the slot
#pragma unroll
        for (int i = 0; i < TP; i++) {
            const int tile = wv + 3 * i;
            if (TP > 1 && tile >= (fine ? tiles_f : tiles_c)) continue;       // wave-uniform: this wave has no such tile
            coop_shade_tile<false, MLP>(P, L, R, t_list, count, tile, tile < (fine ? tiles_f : tiles_c), (fine ? sl.sig_e + fine_e0 : sl.sig_e), lane, wv, col[i], st, sp_direct,
                                        i == 0, depth0);
        }
        return true;
    };
    auto accumulate = [&](int r, bool valid, const v4f (&cc)[TP][2], const v4f (&cf)[TP][2]) {
        if (!valid) return;                 // (what shade(r) returned: no LDS round trip for the ray id)
        PipeSlot sl = pipe_slot<TP>(slots, r & (kPipeSlots - 1));
        const int j = lane & 15, g = lane >> 4;
        // one explicit FMA chain per colour: left to the compiler's contraction, `acc += v0 c0 + v1 c1 + ...` rounds differently in
        // the instantiation with compile-time tile counts (acc known to be zero at the first tile) -- 1 ulp between FULL and not
        float acc[2] = {0.f, 0.f};
#pragma unroll
        for (int i = 0; i < TP; i++) {
            const int tile = wv + 3 * i;
            if (tile < tiles_c) {
                const v4f v = *reinterpret_cast<const v4f*>(sl.v_e + 16 * tile + 4 * g);
#pragma unroll
                for (int n = 0; n < 2; n++)
#pragma unroll
                    for (int k = 0; k < 4; k++) acc[n] = __fmaf_rn(v[k], cc[i][n][k], acc[n]);
            }
            if (tile < tiles_f) {
                const v4f v = *reinterpret_cast<const v4f*>(sl.v_e + fine_e0 + 16 * tile + 4 * g);
#pragma unroll
                for (int n = 0; n < 2; n++)
#pragma unroll
                    for (int k = 0; k < 4; k++) acc[n] = __fmaf_rn(v[k], cf[i][n][k], acc[n]);
            }
        }
#pragma unroll
        for (int n = 0; n < 2; n++) acc[n] = row_sum4(acc[n]);
        if (g < 2) sl.part[wv * 32 + 16 * g + j] = g == 0 ? acc[0] : acc[1];
    };

    auto emit_q = [&](int r, bool fine, const v4f (&col)[TP][2]) {        // BWD: q = sum_c G[c] colour[c] of the samples just shaded -> v_e
        if (r < 0 || r >= nr) return;
        PipeSlot sl = pipe_slot<TP>(slots, r & (kPipeSlots - 1));
        if (__float_as_int(sl.misc[kMiscRay]) < 0) return;
        const int j = lane & 15, g = lane >> 4;
        const float G0 = sl.part[j], G1 = sl.part[16 + j];
#pragma unroll
        for (int i = 0; i < TP; i++) {
            const int tile = wv + 3 * i;
            if (tile >= (fine ? tiles_f : tiles_c)) continue;
            v4f q;
#pragma unroll
            for (int k = 0; k < 4; k++) q[k] = row_total(G0 * col[i][0][k] + G1 * col[i][1][k]);
            if (j == 15) *reinterpret_cast<v4f*>(sl.v_e + (fine ? fine_e0 : 0) + 16 * tile + 4 * g) = q;
        }
    };

    // ------------------------------------------------------------------ the pipeline
    // The scalar wave is one instruction stream against three, shares its SIMD's issue port with MFMA-heavy shader
    // waves, and every step ends when it does: give it issue priority (costs the shaders little, it is mostly waiting
    // on LDS round trips).
    if (wv == 3) __builtin_amdgcn_s_setprio(GNERF_SCALAR_PRIO);
#if GNERF_PIPE_FLAGS && GNERF_PIPE_ROTATE
    if (tid < 4) reinterpret_cast<int*>(unit_rays + kPipeUnit * 8)[tid] = 0;
#endif
    __syncthreads();                                            // weights are in LDS
    if (wv == 3) { propose_issue(0); propose_finish(0); }
    __syncthreads();
    st.reset();
#ifdef GNERF_STAMPS
    // clock of the timed part: shader cycles (s_memtime) per 100 MHz tick (s_memrealtime), and where the wave sits (HW_ID)
    const unsigned long long clk_c0 = __builtin_amdgcn_s_memtime(), clk_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    // Two loops with the same barrier count, one per role.  The roles never change, and with ONE loop and `if (wv < 3)` inside
    // every step the compiler must assume that a wave can enter the scalar branch with the shader branch's registers live (the
    // 32 colour registers) and the other way round: splitting the loops took pipe<2> from 242 to 174 VGPRs and pipe<1> from
    // 0.618-0.635 to 0.60 ms at config 2.
    // (Measured on top of this and dropped: issuing a tile's tap records and its first twelve texel loads BEFORE the barrier that
    // precedes it, with depth proposals one ray earlier and five ray slots, so that the L2 round trip runs under the barrier
    // wait -- 0.617 ms: the two other waves of the SIMD already cover that latency, and the loads' registers are then live across
    // the barrier; its first version spilled two of the loaded vectors, i.e. waited for them on the spot: 0.71 ms.)
    if constexpr (BWD) {
        if (wv < 3) {
            v4f cc[TP][2] = {}, cf[TP][2] = {};
            for (int k = -1; k <= nr + 1; k++) {
                shade(k + 1, false, cc);
                emit_q(k + 1, false, cc);
                __syncthreads();
                shade(k, true, cf);
                emit_q(k, true, cf);
                __syncthreads();
            }
        } else {
            for (int k = -1; k <= nr + 1; k++) {
                finalize(k - 1);
                __syncthreads();
                propose_issue(k + 2);
                importance(k + 1);
                propose_finish(k + 2);
                __syncthreads();
            }
        }
        return;
    }
#if GNERF_PIPE_ROTATE
    // Round 6: the three coarse colour sets change ROLES instead of registers.  Ray k+1's coarse colours are produced in step 2k+2 and
    // consumed in step 2k+7, so three sets are live; the loop used to shift them along every iteration (cc0 = cc1, cc1 = cc2: 16
    // v_mov_b32 at the bottom and 8 at the top, where the compiler parks the loop-carried set -- 72 vector instructions per ray over the
    // three waves).  Unrolled three times, iteration k + i writes set (i + 2) % 3 and reads set i % 3: no copies.  Both roles run
    // the same number of iterations, rounded up to a multiple of three (the extra ones find no ray and only meet at the barriers).
    const int k_last = -1 + 3 * ((nr + 3 + 2) / 3) - 1;
#if GNERF_PIPE_FLAGS
    // progress[0..2]: half-steps the shader waves have finished; progress[3]: the scalar wave's.  One lane writes, every lane of a waiting
    // wave reads the same word(s).  LDS operations of a wave are performed in issue order and the LDS serves one request at a time, so a
    // counter written after a wave's data is seen after that data, and data read before a counter is written has been read by then.
    volatile int* const progress = reinterpret_cast<volatile int*>(unit_rays + kPipeUnit * 8);
    int my_steps = 0;
    auto step_done_shader = [&]() {
        my_steps++;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) progress[wv] = my_steps;
        // go on once the scalar wave has finished the same half-step
        while (__builtin_amdgcn_readfirstlane(progress[3]) < my_steps) __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
    };
    auto step_done_scalar = [&]() {
        my_steps++;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) progress[3] = my_steps;
        while (true) {
            const int a0 = progress[0], a1 = progress[1], a2 = progress[2];
            if (__builtin_amdgcn_readfirstlane(min(a0, min(a1, a2))) >= my_steps) break;
            __builtin_amdgcn_s_sleep(1);
        }
        asm volatile("" ::: "memory");
    };
#define GNERF_STEP_SYNC_SHADER() step_done_shader()
#define GNERF_STEP_SYNC_SCALAR() step_done_scalar()
#else
#define GNERF_STEP_SYNC_SHADER() __syncthreads()
#define GNERF_STEP_SYNC_SCALAR() __syncthreads()
#endif
    if (wv < 3) {
        v4f ca[TP][2] = {}, cb[TP][2] = {}, cd[TP][2] = {}, cf[TP][2] = {};
        bool la = false, lb = false, ld = false;
        auto iter = [&](int k, v4f (&c0)[TP][2], v4f (&c2)[TP][2], bool l0, bool& l2) {
            l2 = shade(k + 1, false, c2);
            GNERF_STAMP(st, 5);
            GNERF_STEP_SYNC_SHADER();
            GNERF_STAMP(st, 6);         // barrier wait, even step
            accumulate(k - 1, l0, c0, cf);
            GNERF_STAMP(st, 10);        // colour accumulate
            shade(k, true, cf);
            GNERF_STAMP(st, 5);
            GNERF_STEP_SYNC_SHADER();
            GNERF_STAMP(st, 7);         // barrier wait, odd step
        };
        for (int k = -1; k <= k_last; k += 3) {         // sets (k-1, k, k+1) = (a, b, d), then (b, d, a), then (d, a, b)
            iter(k, ca, cd, la, ld);
            iter(k + 1, cb, ca, lb, la);
            iter(k + 2, cd, cb, ld, lb);
        }
    } else {
        for (int k = -1; k <= k_last; k++) {
            finalize(k - 1);
            GNERF_STAMP(st, 8);         // merge + final march
            output(k - 2);
            GNERF_STAMP(st, 9);         // outputs
            GNERF_STAMP(st, 5);         // rounding
            GNERF_STEP_SYNC_SCALAR();
            GNERF_STAMP(st, 6);         // barrier wait, even step
            propose_issue(k + 2);
            importance(k + 1);
            GNERF_STAMP(st, 11);        // coarse march + importance
            propose_finish(k + 2);
            GNERF_STAMP(st, 12);        // depth proposals
            GNERF_STAMP(st, 5);
            GNERF_STEP_SYNC_SCALAR();
            GNERF_STAMP(st, 7);         // barrier wait, odd step
        }
    }
#undef GNERF_STEP_SYNC_SHADER
#undef GNERF_STEP_SYNC_SCALAR
#else
    if (wv < 3) {
        v4f cc0[TP][2] = {}, cc1[TP][2] = {}, cc2[TP][2] = {}, cf[TP][2] = {};      // coarse colours of rays k-1, k, k+1; fine colours of ray k-1
        bool live0 = false, live1 = false, live2 = false;                            // do rays k-1, k, k+1 exist (from their coarse pass)
        for (int k = -1; k <= nr + 1; k++) {
            // ---- step 2k+2
            live2 = shade(k + 1, false, cc2);
            GNERF_STAMP(st, 5);
            __syncthreads();
            GNERF_STAMP(st, 6);         // barrier wait, even step
            // ---- step 2k+3
            accumulate(k - 1, live0, cc0, cf);
            GNERF_STAMP(st, 10);        // colour accumulate
            shade(k, true, cf);
#pragma unroll
            for (int i = 0; i < TP; i++) {
#pragma unroll
                for (int n = 0; n < 2; n++) { cc0[i][n] = cc1[i][n]; cc1[i][n] = cc2[i][n]; }
            }
            live0 = live1; live1 = live2;
            GNERF_STAMP(st, 5);
            __syncthreads();
            GNERF_STAMP(st, 7);         // barrier wait, odd step
        }
    } else {
        for (int k = -1; k <= nr + 1; k++) {
            // ---- step 2k+2
            finalize(k - 1);
            GNERF_STAMP(st, 8);         // merge + final march
            output(k - 2);
            GNERF_STAMP(st, 9);         // outputs
            GNERF_STAMP(st, 5);         // rounding
            __syncthreads();
            GNERF_STAMP(st, 6);         // barrier wait, even step
            // ---- step 2k+3
            propose_issue(k + 2);
            importance(k + 1);
            GNERF_STAMP(st, 11);        // coarse march + importance
            propose_finish(k + 2);
            GNERF_STAMP(st, 12);        // depth proposals
            GNERF_STAMP(st, 5);
            __syncthreads();
            GNERF_STAMP(st, 7);         // barrier wait, odd step
        }
    }
#endif
#ifdef GNERF_STAMPS
    if (lane == 0 && p.debug) {
        unsigned long long* out = reinterpret_cast<unsigned long long*>(p.debug) + (size_t(blockIdx.x) * 4 + wv) * 16;
        for (int i = 0; i < 16; i++) out[i] = st.acc[i];
        out[15] = nr;
        unsigned long long* ext = reinterpret_cast<unsigned long long*>(p.debug) + size_t(gridDim.x) * 4 * 16 + (size_t(blockIdx.x) * 4 + wv) * 4;
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        ext[0] = hw | ((unsigned long long)(xcc & 15u) << 32);
        ext[1] = __builtin_amdgcn_s_memtime() - clk_c0;
        ext[2] = __builtin_amdgcn_s_memrealtime() - clk_r0;
        ext[3] = clk_r0;
    }
#endif
    if (wv == 3 && lane == 0) range.flush(P);
}

template <int TP, int MLP, bool FULL, bool GEN = false>
__global__ __launch_bounds__(kPipeThreads, TP == 1 ? GNERF_PIPE_WAVES_PER_SIMD : (TP == 2 ? GNERF_PIPE2_WAVES_PER_SIMD : 2)) void render_kernel_pipe(Params P) {
    extern __shared__ __align__(16) float smem[];
    if constexpr (MLP == kMlpAuto) {            // see render_kernel_coop
        bool sp_direct;
        if (choose_mlp(P, smem, &sp_direct) == kMlpF32) render_pipe_body<TP, kMlpF32, FULL, GEN>(P, smem);
        else                                           render_pipe_body<TP, kMlpF16x3, FULL, GEN>(P, smem, nullptr, nullptr, sp_direct);
    } else {
        render_pipe_body<TP, MLP, FULL, GEN>(P, smem);
    }
}

// First pass of the staged backward: the forward pipeline in its BWD form (see render_pipe_body).  General sample counts (FULL = false:
// any S, F the pipelined kernels cover, disparity sampling, per-ray limits), decoder arithmetic chosen on the device as in the forward.
template <int TP>
__global__ __launch_bounds__(kPipeThreads, TP == 1 ? GNERF_PIPE_WAVES_PER_SIMD : (TP == 2 ? GNERF_PIPE2_WAVES_PER_SIMD : 2)) void render_kernel_pipe_bwd(Params P, gnerf_render_grads Gr, float* stage) {
    extern __shared__ __align__(16) float smem[];
    int mlp = P.p.mlp_mode;                     // (GNERF_BWD_MLP forces one; the launcher passes AUTO otherwise)
    bool sp_direct = false;
    if (mlp == kMlpAuto) mlp = choose_mlp(P, smem, &sp_direct);
    if (mlp == kMlpF32) render_pipe_body<TP, kMlpF32, false, false, true>(P, smem, &Gr, stage);
    else                render_pipe_body<TP, kMlpF16x3, false, false, true>(P, smem, &Gr, stage, sp_direct);
}
