// Backward pass of the fused renderer.  Included by render.hip (inside its anonymous namespace).
//
// Upstream this is autograd walking the graph of ImportanceRenderer.forward (renderer.py:88-140) backwards:
// the composite of ray_marcher.py:25-57, the sort/gather of renderer.py:150-167, the decoder's two linear layers
// (triplane.py:113-136) and grid_sample's backward (grid_sample_gradfix.py:62-77), with ~4 GB of saved
// intermediates at the training shape.  Here nothing is saved: ONE WAVE OWNS A RAY and
//   1. recomputes the forward pass of its ray (same depth proposals, lookups, MLP, coarse march, importance
//      resampling, merge, final march) keeping only per-sample scalars in LDS -- in place of each sample's
//      32 colours it keeps q_j = sum_c dL/drgb[c] * colour_j[c], which is all the composite's gradient needs;
//   2. runs the composite's gradient as wave scans over the sorted intervals: dL/dw_k, the suffix sums that
//      carry the transmittance dependence, dL/dsigma of every sample and the colour weights v_j;
//   3. walks the sample tiles once more: lookup + MLP forward again, then dO -> dH -> dPRE -> dX on the matrix
//      cores (exact fp32 v_mfma_f32_16x16x4_f32; operands change roles through small LDS transposes), the
//      weight gradients dW2 += dO^T H and dW1 += dPRE^T X accumulated in registers across all rays of the wave,
//      and dX scattered into the plane gradient with one hardware float atomic per (tap, channel).
// Four waves (four rays) share a workgroup and one LDS copy of the decoder; the weight gradients are reduced
// in LDS per workgroup and leave with one atomic per element.
//
// Conventions as in render.hip: MFMA D layout col = lane & 15, rows = 4 * (lane >> 4) + r.

constexpr int kBwdWaves = 4;
constexpr int kBwdThreads = 64 * kBwdWaves;
constexpr int kBwdRaysPerWave = 16;
constexpr int kTPitch = 36;          // dO [16 samples][32 colour outputs], later dX [16][32 channels]
constexpr int kHPitch = 68;          // H, later dPRE: [16 samples][64 hidden]
constexpr int kBwdWeightFloats = 64 * kW1Pitch + 33 * kW2Pitch + 64 + 36;      // w1, w2, b1, b2 in plain fp32 rows
constexpr int kBwdGradFloats = 64 * 32 + 64 + 33 * 64 + 33;                    // reduction area, aliases the weights
static_assert(kBwdGradFloats <= kBwdWeightFloats, "weight-gradient reduction area must fit in the weight area");

__host__ __device__ inline size_t bwd_wave_floats(int s_pad) {
    return size_t(12) * s_pad + 16 * kStagePitch + 16 * kTPitch + 16 * kHPitch + 16 * kTapDwords;
}

struct BwdLds {
    const float* w1; const float* w2; const float* b1; const float* b2;     // workgroup-shared decoder
    float* t_e; float* sig_e; float* q_e; float* v_e; float* dsig_e; int* rank_e;      // per element (coarse k at k, fine i at 16*tiles_c + i)
    float* s_t; float* s_sig; float* s_q; float* w_s; float* trans; float* ds;         // sorted order / per interval
    float* stage; float* tbuf; float* hbuf; float* taps;
};


#ifdef GNERF_ABLATE_ATOMIC       // timing-only build: plain stores instead of atomics (wrong results)
#define GNERF_SCATTER_ADD(ptr, val) (*(ptr) = (val))
#else
#define GNERF_SCATTER_ADD(ptr, val) unsafeAtomicAdd(ptr, val)
#endif

struct BwdRay { float ox, oy, oz, dx, dy, dz; const char* planes; };

// sample j of tile `tile` of a ray: depth from t_list (clamped to the last real sample for tile padding)
struct RayTilePos {
    const BwdRay& R; const float* t_list; int count, tile; float box_scale;
    __device__ __forceinline__ void operator()(int j, float& px, float& py, float& pz) const {
        const float depth = t_list[min(16 * tile + j, count - 1)];
        px = __fadd_rn(R.ox, __fmul_rn(depth, R.dx)) * box_scale;
        py = __fadd_rn(R.oy, __fmul_rn(depth, R.dy)) * box_scale;
        pz = __fadd_rn(R.oz, __fmul_rn(depth, R.dz)) * box_scale;
    }
};

// A tile's plane-gradient scatter that has not been issued yet: dX[16][32] sits in tbuf, its tap records in taps.
// It is issued from inside the NEXT tile's lookup, after that tile's texel loads: the wave then waits only for the
// loads, and the atomics drain while the next tile's MLP runs.  (Issued right after dX is produced, the next lookup's
// s_waitcnt vmcnt would wait for every atomic to be acknowledged by L2 first: atomic and compute time simply add up.)
struct BwdPending { float* base; int live; };

// One sixth of a pending scatter: samples 8a..8a+7 of plane pl; lane = (sample parity, channel).
__device__ __forceinline__ void bwd_scatter_chunk(const BwdLds& L, const BwdPending& pd, int a, int pl, int lane) {
    const int half = lane >> 5, ch = lane & 31;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int smp = 8 * a + 2 * q + half;
        const float val = smp < pd.live ? L.tbuf[smp * kTPitch + ch] : 0.f;
        if (val != 0.f) {                   // exact zeros (padding, samples behind an opaque surface) cost no atomic
            const float* rec = L.taps + smp * kTapDwords + pl * 8;
            const uint4 off = *reinterpret_cast<const uint4*>(rec);
            const v4f wgt = *reinterpret_cast<const v4f*>(rec + 4);
            if (wgt[0] != 0.f) GNERF_SCATTER_ADD(pd.base + (off.x >> 2) + ch, val * wgt[0]);
            if (wgt[1] != 0.f) GNERF_SCATTER_ADD(pd.base + (off.y >> 2) + ch, val * wgt[1]);
            if (wgt[2] != 0.f) GNERF_SCATTER_ADD(pd.base + (off.z >> 2) + ch, val * wgt[2]);
            if (wgt[3] != 0.f) GNERF_SCATTER_ADD(pd.base + (off.w >> 2) + ch, val * wgt[3]);
        }
    }
}

// Tap records of the 16 samples of a tile (lanes 0..47: sample = lane & 15, plane = lane >> 4), then the lookup:
// 8 lanes per texel, 8 samples per step, blended features staged as X[sample][channel].  The records travel through
// hbuf (free between tiles) because `taps` may still hold the pending scatter's records; with KEEP_TAPS they are
// written to `taps` once the pending scatter has been issued.
// `pos(j, px, py, pz)` yields the position of sample j of the tile, already scaled into the planes' [-1,1] frame.
template <bool KEEP_TAPS, class PosFn, bool SCATTER = true>
__device__ __forceinline__ void bwd_gather_tile(const Params& P, const BwdLds& L, const char* planes, PosFn pos, BwdPending& pend, int lane) {
    const int H = P.p.plane_h, W = P.p.plane_w;
    uint4 my_off = make_uint4(0, 0, 0, 0);
    v4f my_wgt = {0.f, 0.f, 0.f, 0.f};
    if (lane < 48) {
        const int j = lane & 15, pl = lane >> 4;
        float px, py, pz;
        pos(j, px, py, pz);
        const float u = pl == 2 ? pz : px;
        const float v = pl == 0 ? py : (pl == 1 ? pz : px);
        plane_taps(H, W, u, v, P.tex_pitch, P.row_pitch, unsigned(pl) * P.plane_pitch, my_off, my_wgt);
        float* rec = L.hbuf + j * kTapDwords + pl * 8;
        *reinterpret_cast<uint4*>(rec) = my_off;
        *reinterpret_cast<v4f*>(rec + 4) = my_wgt;
    }
    lds_wave_sync();
    const int b = lane >> 3, cq16 = (lane & 7) * 16;
#pragma unroll
    for (int a = 0; a < 2; a++) {
        const int js = 8 * a + b;
        v4f acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int pl = 0; pl < 3; pl++) {
            const float* rec = L.hbuf + js * kTapDwords + pl * 8;
            const uint4 off = *reinterpret_cast<const uint4*>(rec);
            const v4f wgt = *reinterpret_cast<const v4f*>(rec + 4);
            const v4f t00 = *reinterpret_cast<const v4f*>(planes + off.x + cq16);
            const v4f t01 = *reinterpret_cast<const v4f*>(planes + off.y + cq16);
            const v4f t10 = *reinterpret_cast<const v4f*>(planes + off.z + cq16);
            const v4f t11 = *reinterpret_cast<const v4f*>(planes + off.w + cq16);
            if constexpr (SCATTER) { if (pend.live > 0) bwd_scatter_chunk(L, pend, a, pl, lane); }
            acc += t00 * wgt[0] + t01 * wgt[1] + t10 * wgt[2] + t11 * wgt[3];
        }
        *reinterpret_cast<v4f*>(L.stage + js * kStagePitch + (cq16 >> 2)) = acc;
    }
    pend.live = 0;
    if (KEEP_TAPS && lane < 48) {
        float* rec = L.taps + (lane & 15) * kTapDwords + (lane >> 4) * 8;
        *reinterpret_cast<uint4*>(rec) = my_off;
        *reinterpret_cast<v4f*>(rec + 4) = my_wgt;
    }
    lds_wave_sync();
}

// The same lookup for kernels that hold no scatter state (render_bwd_tiles_kernel): the tile's 24 texel loads run as the forward's ROLLING
// window (coop_shade_tile, render_coop.inl) -- the three planes of step 0 in flight together, and as soon as a plane of step 0 has been
// blended the same plane of step 1 is issued -- instead of six dependent rounds of four loads: with two waves per SIMD a round trip to
// L2 per round was a third of the tile's time.
template <class PosFn>
__device__ __forceinline__ void bwd_gather_tile_rolling(const Params& P, const BwdLds& L, const char* planes, PosFn pos, int lane) {
    const int H = P.p.plane_h, W = P.p.plane_w;
    if (lane < 48) {
        const int j = lane & 15, pl = lane >> 4;
        float px, py, pz;
        pos(j, px, py, pz);
        const float u = pl == 2 ? pz : px;
        const float v = pl == 0 ? py : (pl == 1 ? pz : px);
        uint4 my_off; v4f my_wgt;
        plane_taps(H, W, u, v, P.tex_pitch, P.row_pitch, unsigned(pl) * P.plane_pitch, my_off, my_wgt);
        float* rec = L.hbuf + j * kTapDwords + pl * 8;
        *reinterpret_cast<uint4*>(rec) = my_off;
        *reinterpret_cast<v4f*>(rec + 4) = my_wgt;
    }
    lds_wave_sync();
    const int b = lane >> 3, cq16 = (lane & 7) * 16;
    uint4 off[2][3];
    v4f wgt[2][3], tex[2][3][4];
    auto read_records = [&](int a) {
        const float* rec = L.hbuf + (8 * a + b) * kTapDwords;
#pragma unroll
        for (int pl = 0; pl < 3; pl++) {
            off[a][pl] = *reinterpret_cast<const uint4*>(rec + pl * 8);
            wgt[a][pl] = *reinterpret_cast<const v4f*>(rec + pl * 8 + 4);
        }
    };
    auto issue = [&](int a, int pl) {
#ifdef GNERF_ABLATE_K2GATHER     // timing-only build: no texel loads (outputs are wrong)
        tex[a][pl][0] = (v4f){float(off[a][pl].x + cq16), 1.f, 2.f, 3.f}; tex[a][pl][1] = (v4f){float(off[a][pl].y), 1.f, 2.f, 3.f};
        tex[a][pl][2] = (v4f){float(off[a][pl].z), 1.f, 2.f, 3.f};        tex[a][pl][3] = (v4f){float(off[a][pl].w), 1.f, 2.f, 3.f};
        return;
#endif
        tex[a][pl][0] = *reinterpret_cast<const v4f*>(planes + (off[a][pl].x + cq16));
        tex[a][pl][1] = *reinterpret_cast<const v4f*>(planes + (off[a][pl].y + cq16));
        tex[a][pl][2] = *reinterpret_cast<const v4f*>(planes + (off[a][pl].z + cq16));
        tex[a][pl][3] = *reinterpret_cast<const v4f*>(planes + (off[a][pl].w + cq16));
    };
    auto blend = [&](int a, int pl, v4f& acc) {                 // the forward's chain (coop_shade_tile): same features, same bits
        auto bc = [](float w) { return (v4f){w, w, w, w}; };
        acc = pl == 0 ? tex[a][pl][0] * wgt[a][pl][0] : __builtin_elementwise_fma(tex[a][pl][0], bc(wgt[a][pl][0]), acc);
        acc = __builtin_elementwise_fma(tex[a][pl][1], bc(wgt[a][pl][1]), acc);
        acc = __builtin_elementwise_fma(tex[a][pl][2], bc(wgt[a][pl][2]), acc);
        acc = __builtin_elementwise_fma(tex[a][pl][3], bc(wgt[a][pl][3]), acc);
    };
    v4f acc0, acc1;
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
    *reinterpret_cast<v4f*>(L.stage + b * kStagePitch + (lane & 7) * 4) = acc0;
    blend(1, 0, acc1); blend(1, 1, acc1); blend(1, 2, acc1);
    *reinterpret_cast<v4f*>(L.stage + (8 + b) * kStagePitch + (lane & 7) * 4) = acc1;
    lds_wave_sync();
}

// MLP forward on the staged tile: h = softplus(W1 x + b1) in the H^T layout (lane: sample j, hidden 16m+4g+r),
// o = colour pre-activations (lane: sample 4g+r, output 1+16n+j), sig = density of sample j.
__device__ __forceinline__ void bwd_mlp_forward(const BwdLds& L, int lane, v4f (&h)[4], v4f (&o)[2], float& sig) {
    const int j = lane & 15, g = lane >> 4;
    const v4f f_lo = *reinterpret_cast<const v4f*>(L.stage + j * kStagePitch + 8 * g);
    const v4f f_hi = *reinterpret_cast<const v4f*>(L.stage + j * kStagePitch + 8 * g + 4);
    const float f[8] = {f_lo[0], f_lo[1], f_lo[2], f_lo[3], f_hi[0], f_hi[1], f_hi[2], f_hi[3]};
#pragma unroll
    for (int m = 0; m < 4; m++) {
        h[m] = *reinterpret_cast<const v4f*>(L.b1 + 16 * m + 4 * g);
        const v4f a_lo = *reinterpret_cast<const v4f*>(L.w1 + (16 * m + j) * kW1Pitch + 8 * g);
        const v4f a_hi = *reinterpret_cast<const v4f*>(L.w1 + (16 * m + j) * kW1Pitch + 8 * g + 4);
#pragma unroll
        for (int s = 0; s < 4; s++) h[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_lo[s], f[s], h[m], 0, 0, 0);
#pragma unroll
        for (int s = 0; s < 4; s++) h[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_hi[s], f[4 + s], h[m], 0, 0, 0);
    }
    sig = 0.f;
#pragma unroll
    for (int m = 0; m < 4; m++) {
        const v4f ws = *reinterpret_cast<const v4f*>(L.w2 + 16 * m + 4 * g);          // density row W2[0][:]
#pragma unroll
        for (int r = 0; r < 4; r++) {
            h[m][r] = softplus_hw(h[m][r]);
            sig += ws[r] * h[m][r];
        }
    }
    sig += __shfl_xor(sig, 16);
    sig += __shfl_xor(sig, 32);
    sig += L.b2[0];
#pragma unroll
    for (int n = 0; n < 2; n++) {
        const float bias = L.b2[1 + 16 * n + j];
        o[n] = (v4f){bias, bias, bias, bias};
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const v4f bw = *reinterpret_cast<const v4f*>(L.w2 + (1 + 16 * n + j) * kW2Pitch + 16 * m + 4 * g);
#pragma unroll
            for (int r = 0; r < 4; r++) o[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(h[m][r], bw[r], o[n], 0, 0, 0);
        }
    }
}

// Forward-only walk of `ntiles` tiles: densities -> sig_e[e0...], q -> q_e[e0...].
// G[n] = dL/d(colour sum) of channel 16n + (lane & 15), i.e. 2 * grad_rgb.
template <bool STAGED>
__device__ __forceinline__ void bwd_forward_tiles(const Params& P, const BwdLds& L, const BwdRay& R, int e0, int count, int ntiles,
                                                  const float (&G)[2], BwdPending& pend, int lane) {
    const int j = lane & 15, g = lane >> 4;
    for (int t = 0; t < ntiles; t++) {
        bwd_gather_tile<false, RayTilePos, !STAGED>(P, L, R.planes, RayTilePos{R, L.t_e + e0, count, t, P.box_scale}, pend, lane);
        v4f h[4], o[2];
        float sig;
        bwd_mlp_forward(L, lane, h, o, sig);
        if (g == 0 && 16 * t + j < count) L.sig_e[e0 + 16 * t + j] = sig;
        v4f q;
#pragma unroll
        for (int r = 0; r < 4; r++) q[r] = row_total(G[0] * sigmoid_rgb_hw(o[0][r]) + G[1] * sigmoid_rgb_hw(o[1][r]));
        if (j == 15) *reinterpret_cast<v4f*>(L.q_e + e0 + 16 * t + 4 * g) = q;
    }
    lds_wave_sync();
}

// Per-wave accumulators of the decoder gradients (summed over every sample the wave shades).
struct BwdAcc {
    v4f w1[4][2];     // dW1[16m + 4g + r][16c + j]
    v4f w2[2][4];     // dW2[1 + 16o + 4g + r][16n + j]
    v4f w2s[4];       // dW2[0][16m + 4g + r], partial over the samples on this lane's column
    v4f b1[4];        // db1[16m + 4g + r], partial likewise
    float b2[2];      // db2[1 + 16n + j], partial over this lane's sample rows
    float b2s;        // db2[0], partial
};

// The decoder's backward pass for one staged tile.  On entry: X in `stage`, dO[16][32] (colour pre-activation gradients)
// in `tbuf`, h = the hidden activations (H^T layout), dsig = dL/dsigma of this lane's sample.  Accumulates the weight
// gradients into A and leaves dX[16][32] in `tbuf`.
__device__ __forceinline__ void bwd_tile_core(const BwdLds& L, v4f (&h)[4], float dsig, BwdAcc& A, int lane) {
    const int j = lane & 15, g = lane >> 4;
    if (g == 0) A.b2s += dsig;
    lds_wave_sync();
    // ---- dH^T[hidden][sample] = sum_out W2[out][hidden] dO[sample][out]  (+ the density row on the vector ALU)
    v4f dh[4];
#pragma unroll
    for (int m = 0; m < 4; m++) {
        const v4f ws = *reinterpret_cast<const v4f*>(L.w2 + 16 * m + 4 * g);
        dh[m] = ws * dsig;
        A.w2s[m] += h[m] * dsig;
    }
#pragma unroll
    for (int s = 0; s < 8; s++) {
        const float bT = L.tbuf[j * kTPitch + 4 * s + g];
#pragma unroll
        for (int m = 0; m < 4; m++)
            dh[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(L.w2[(1 + 4 * s + g) * kW2Pitch + 16 * m + j], bT, dh[m], 0, 0, 0);
    }
    // ---- through softplus: d/dpre softplus(pre) = 1 - exp(-softplus(pre))
#pragma unroll
    for (int m = 0; m < 4; m++) {
#pragma unroll
        for (int r = 0; r < 4; r++) dh[m][r] *= 1.f - exp_hw(-h[m][r]);
        A.b1[m] += dh[m];
        *reinterpret_cast<v4f*>(L.hbuf + j * kHPitch + 16 * m + 4 * g) = h[m];
    }
    lds_wave_sync();
    // ---- dW2[out][hidden] += sum_sample dO[sample][out] H[sample][hidden]
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const float a0 = L.tbuf[(4 * g + r) * kTPitch + j], a1 = L.tbuf[(4 * g + r) * kTPitch + 16 + j];
#pragma unroll
        for (int n = 0; n < 4; n++) {
            const float bh = L.hbuf[(4 * g + r) * kHPitch + 16 * n + j];
            A.w2[0][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bh, A.w2[0][n], 0, 0, 0);
            A.w2[1][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bh, A.w2[1][n], 0, 0, 0);
        }
    }
    lds_wave_sync();
    // ---- dX^T[channel][sample] = sum_hidden W1[hidden][channel] dPRE^T[hidden][sample]
    v4f dx[2];
    dx[0] = dx[1] = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < 4; m++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const float* row = L.w1 + (16 * m + 4 * g + r) * kW1Pitch + j;
            dx[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(row[0], dh[m][r], dx[0], 0, 0, 0);
            dx[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(row[16], dh[m][r], dx[1], 0, 0, 0);
        }
        *reinterpret_cast<v4f*>(L.hbuf + j * kHPitch + 16 * m + 4 * g) = dh[m];             // dPRE[sample][hidden]
    }
    *reinterpret_cast<v4f*>(L.tbuf + j * kTPitch + 4 * g) = dx[0];                            // dX[sample][channel]
    *reinterpret_cast<v4f*>(L.tbuf + j * kTPitch + 16 + 4 * g) = dx[1];
    lds_wave_sync();
    // ---- dW1[hidden][channel] += sum_sample dPRE[sample][hidden] X[sample][channel]
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const float b0 = L.stage[(4 * g + r) * kStagePitch + j], b1 = L.stage[(4 * g + r) * kStagePitch + 16 + j];
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const float ap = L.hbuf[(4 * g + r) * kHPitch + 16 * m + j];
            A.w1[m][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap, b0, A.w1[m][0], 0, 0, 0);
            A.w1[m][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap, b1, A.w1[m][1], 0, 0, 0);
        }
    }
}

// `stage_ray`: when not NULL the plane gradient is NOT scattered from here: each sample's dX row (32 floats) is written to
// stage_ray[rank * 32 ...], rank = the sample's position in the ray's merged depth order, for plane_scatter_kernel below.
template <bool STAGED>
__device__ __forceinline__ void bwd_backward_tiles(const Params& P, const BwdLds& L, const BwdRay& R, float* grad_planes_item, float* stage_ray,
                                                   int e0, int count, int ntiles, const float (&G)[2], BwdAcc& A, BwdPending& pend, int lane) {
    const int j = lane & 15, g = lane >> 4;
    for (int t = 0; t < ntiles; t++) {
        bwd_gather_tile<!STAGED, RayTilePos, !STAGED>(P, L, R.planes, RayTilePos{R, L.t_e + e0, count, t, P.box_scale}, pend, lane);
#ifdef GNERF_ABLATE_BWDMLP      // timing-only build: lookup + scatter without the decoder's backward pass (wrong results)
        for (int i = lane; i < 16 * kTPitch; i += 64) L.tbuf[i] = L.stage[i];
        if (grad_planes_item) { pend.base = grad_planes_item; pend.live = min(16, count - 16 * t); }
        lds_wave_sync();
        continue;
#endif
        v4f h[4], o[2];
        float sig;
        bwd_mlp_forward(L, lane, h, o, sig);
        // ---- dO: colour c = 1.002 * s - 0.001 with s = sigmoid(o); dL/dc = G * v_sample  (triplane.py:134, ray_marcher.py:27-45)
        const v4f vs = *reinterpret_cast<const v4f*>(L.v_e + e0 + 16 * t + 4 * g);
        const float dsig = L.dsig_e[e0 + 16 * t + j];
#pragma unroll
        for (int n = 0; n < 2; n++) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const float e = __builtin_amdgcn_exp2f(o[n][r] * -1.44269504088896341f);
                const float s = __builtin_amdgcn_rcpf(1.0f + e);
                const float d = G[n] * vs[r] * (1.002f * s * (1.f - s));
                L.tbuf[(4 * g + r) * kTPitch + 16 * n + j] = d;
                A.b2[n] += d;
            }
        }
        bwd_tile_core(L, h, dsig, A, lane);
        if constexpr (STAGED) {
            // ---- dX rows to the staging buffer in depth order: two 128-byte rows per store instruction
            const int half = lane >> 5, ch = lane & 31;
            const int live = min(16, count - 16 * t);
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const int smp = 2 * q + half;
                if (smp < live) stage_ray[L.rank_e[e0 + 16 * t + smp] * 32 + ch] = L.tbuf[smp * kTPitch + ch];
            }
        } else if (grad_planes_item) {
            // ---- dX goes to the plane gradient from inside the next lookup (see BwdPending)
            pend.base = grad_planes_item; pend.live = min(16, count - 16 * t);
        }
        lds_wave_sync();
    }
}

// Decoder gradients: wave registers -> workgroup LDS (aliases the staged decoder) -> one global atomic per element.
// Every thread of the workgroup must call (barriers inside).
__device__ __forceinline__ void bwd_reduce_decoder_grads(const BwdAcc& A, float* smem, float* grad_w1, float* grad_b1, float* grad_w2, float* grad_b2,
                                                         int tid, int lane, int j, int nthreads = kBwdThreads) {
    if (!grad_w1) return;
    __syncthreads();                                   // every wave is done with the LDS decoder
    float* red = smem;
    float* red_w1 = red, *red_b1 = red_w1 + 64 * 32, *red_w2 = red_b1 + 64, *red_b2 = red_w2 + 33 * 64;
    for (int i = tid; i < kBwdGradFloats; i += nthreads) red[i] = 0.f;
    __syncthreads();
    {
        const int g = lane >> 4;
#pragma unroll
        for (int m = 0; m < 4; m++) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int hid = 16 * m + 4 * g + r;
                atomicAdd(red_w1 + hid * 32 + j, A.w1[m][0][r]);
                atomicAdd(red_w1 + hid * 32 + 16 + j, A.w1[m][1][r]);
                const float s0 = row_total(A.w2s[m][r]), s1 = row_total(A.b1[m][r]);
                if (j == 15) { atomicAdd(red_w2 + hid, s0); atomicAdd(red_b1 + hid, s1); }
            }
        }
#pragma unroll
        for (int o = 0; o < 2; o++) {
#pragma unroll
            for (int n = 0; n < 4; n++) {
#pragma unroll
                for (int r = 0; r < 4; r++) atomicAdd(red_w2 + (1 + 16 * o + 4 * g + r) * 64 + 16 * n + j, A.w2[o][n][r]);
            }
            float s = A.b2[o];
            s += __shfl_xor(s, 16);
            s += __shfl_xor(s, 32);
            if (g == 0) atomicAdd(red_b2 + 1 + 16 * o + j, s);
        }
        const float s = wave_sum(A.b2s);
        if (lane == 0) atomicAdd(red_b2, s);
    }
    __syncthreads();
    for (int i = tid; i < 64 * 32; i += nthreads) unsafeAtomicAdd(grad_w1 + i, red_w1[i]);
    for (int i = tid; i < 33 * 64; i += nthreads) unsafeAtomicAdd(grad_w2 + i, red_w2[i]);
    if (tid < 64) unsafeAtomicAdd(grad_b1 + tid, red_b1[tid]);
    if (tid < 33) unsafeAtomicAdd(grad_b2 + tid, red_b2[tid]);
}

// STAGED: the plane gradient leaves through the staging buffer (`stage` != NULL, a plane gradient is requested): no tap records are
// kept and no scatter code is compiled in.
#ifndef GNERF_BWD_WAVE_OCC
#define GNERF_BWD_WAVE_OCC 2        // waves per SIMD the one-wave-per-ray kernel is compiled for (1: up to 512 registers, no spills)
#endif
template <bool STAGED>
__global__ __launch_bounds__(kBwdThreads, GNERF_BWD_WAVE_OCC) void render_bwd_kernel(Params P, gnerf_render_grads Gr, float* stage) {
    extern __shared__ __align__(16) float smem[];
    const gnerf_render_params& p = P.p;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int S = p.depth_resolution, F = p.depth_resolution_importance;
    const int s_pad = 16 * (P.tiles_c + P.tiles_f);
    const int n_all = S + F;
    const int fine_e0 = 16 * P.tiles_c;

    // ---- decoder into LDS (plain fp32 rows; every matrix is read in two orientations)
    float* w1 = smem;
    float* w2 = w1 + 64 * kW1Pitch;
    float* b1 = w2 + 33 * kW2Pitch;
    float* b2 = b1 + 64;
    for (int i = tid; i < 64 * 32; i += kBwdThreads) w1[(i >> 5) * kW1Pitch + (i & 31)] = p.w1[i];
    for (int i = tid; i < 33 * 64; i += kBwdThreads) w2[(i >> 6) * kW2Pitch + (i & 63)] = p.w2[i];
    if (tid < 64) b1[tid] = p.b1[tid];
    if (tid < 36) b2[tid] = tid < 33 ? p.b2[tid] : 0.f;
    __syncthreads();

    BwdLds L;
    L.w1 = w1; L.w2 = w2; L.b1 = b1; L.b2 = b2;
    float* base = smem + kBwdWeightFloats + size_t(wv) * bwd_wave_floats(s_pad);
    L.t_e = base;               L.sig_e = L.t_e + s_pad;   L.q_e = L.sig_e + s_pad;  L.v_e = L.q_e + s_pad;
    L.dsig_e = L.v_e + s_pad;   L.rank_e = reinterpret_cast<int*>(L.dsig_e + s_pad);
    L.s_t = L.dsig_e + 2 * s_pad;  L.s_sig = L.s_t + s_pad;  L.s_q = L.s_sig + s_pad;  L.w_s = L.s_q + s_pad;
    L.trans = L.w_s + s_pad;    L.ds = L.trans + s_pad;
    L.stage = L.ds + s_pad;     L.tbuf = L.stage + 16 * kStagePitch;  L.hbuf = L.tbuf + 16 * kTPitch;  L.taps = L.hbuf + 16 * kHPitch;

    BwdAcc A;
#pragma unroll
    for (int m = 0; m < 4; m++) {
        A.w1[m][0] = A.w1[m][1] = A.w2[0][m] = A.w2[1][m] = A.w2s[m] = A.b1[m] = (v4f){0.f, 0.f, 0.f, 0.f};
    }
    A.b2[0] = A.b2[1] = A.b2s = 0.f;
    BwdPending pend = {nullptr, 0};

    // ray tiles: workgroup `blk` owns tiles 4*blk .. 4*blk+3 (one per wave); XCD-contiguous like the forward kernels
    // Small launches (the training shape is 1 024 tiles = one wave per SIMD): a tile's 16 rays are shared by 1 << split_shift waves
    const int n_groups = P.n_tiles << P.split_shift;
    const int n_blocks = (n_groups + kBwdWaves - 1) / kBwdWaves;
    const int per_xcd = (n_blocks + kNumXCD - 1) / kNumXCD;
    const int blk = (blockIdx.x % kNumXCD) * per_xcd + blockIdx.x / kNumXCD;
    const int group = blk * kBwdWaves + wv;
    const int tile = group >> P.split_shift;
    const bool have_tile = blk < n_blocks && group < n_groups;
    const int rr_count = kBwdRaysPerWave >> P.split_shift;
    const int rr_first = (group & ((1 << P.split_shift) - 1)) * rr_count;
    const int64_t plane_floats = int64_t(3) * p.plane_h * p.plane_w * 32;
    const int j = lane & 15;

    for (int rr = rr_first; have_tile && rr < rr_first + rr_count; rr++) {
        int64_t ray;
        if (P.tiles_per_item > 0) {
            const int item = tile / P.tiles_per_item, tt = tile % P.tiles_per_item;
            const int tx = tt / P.tiles_y, ty = tt % P.tiles_y;
            ray = int64_t(item) * p.rays_per_item + int64_t(ty * 4 + (rr >> 2)) * p.image_width + tx * 4 + (rr & 3);
        } else {
            ray = int64_t(tile) * kBwdRaysPerWave + rr;
            if (ray >= P.total_rays) break;
        }
        const int item = int(ray / p.rays_per_item);
        BwdRay R;
        R.ox = p.ray_origins[ray * 3 + 0]; R.oy = p.ray_origins[ray * 3 + 1]; R.oz = p.ray_origins[ray * 3 + 2];
        R.dx = p.ray_dirs[ray * 3 + 0]; R.dy = p.ray_dirs[ray * 3 + 1]; R.dz = p.ray_dirs[ray * 3 + 2];
        R.planes = reinterpret_cast<const char*>(p.planes_nhwc + int64_t(item) * plane_floats);
        float* grad_planes_item = Gr.grad_planes_nhwc ? Gr.grad_planes_nhwc + int64_t(item) * plane_floats : nullptr;
        // incoming gradients; rgb = 2 * composite - 1 (ray_marcher.py:55)
        float G[2];
        G[0] = Gr.grad_rgb ? 2.f * Gr.grad_rgb[ray * 32 + j] : 0.f;
        G[1] = Gr.grad_rgb ? 2.f * Gr.grad_rgb[ray * 32 + 16 + j] : 0.f;
        const float g_depth = Gr.grad_depth ? Gr.grad_depth[ray] : 0.f;
        const float g_wsum = Gr.grad_wsum ? Gr.grad_wsum[ray] : 0.f;
        const float g_sum = wave_sum(lane < 16 ? G[0] + G[1] : 0.f);

        // ---- 1. forward recomputation
        for (int k = lane; k < S; k += 64) L.t_e[k] = coarse_depth(P, ray, k);
        for (int k = lane; k < s_pad; k += 64) { L.v_e[k] = 0.f; L.dsig_e[k] = 0.f; L.q_e[k] = 0.f; }
        lds_wave_sync();
        bwd_forward_tiles<STAGED>(P, L, R, 0, S, P.tiles_c, G, pend, lane);
        if (F > 0) {
            float w_sum, wt_sum;
            march(L.t_e, L.sig_e, L.w_s, S, lane, w_sum, wt_sum);
            lds_wave_sync();
            resample_fine(P, ray, L.t_e, L.w_s, L.s_sig, L.trans, L.t_e + fine_e0, nullptr, n_all, lane, [] { lds_wave_sync(); });
            bwd_forward_tiles<STAGED>(P, L, R, fine_e0, F, P.tiles_f, G, pend, lane);
            merge_by_depth(L.t_e, L.sig_e, L.rank_e, L.s_t, L.s_sig, S, F, fine_e0, lane);
        } else {
            for (int k = lane; k < S; k += 64) { L.rank_e[k] = k; L.s_t[k] = L.t_e[k]; L.s_sig[k] = L.sig_e[k]; }
        }
        lds_wave_sync();
        for (int q = lane; q < n_all; q += 64) {
            const int e = q < S ? q : fine_e0 + (q - S);
            L.s_q[L.rank_e[e]] = L.q_e[e];
        }
        // final march, keeping the transmittance in front of every interval (ray_marcher.py:26-42)
        float w_sum, wt_sum;
        {
            float carry = 1.f, acc_w = 0.f, acc_wt = 0.f;
            for (int kb = 0; kb < n_all - 1; kb += 64) {
                const int k = kb + lane;
                const bool ok = k < n_all - 1;
                float alpha = 0.f, tmid = 0.f;
                if (ok) {
                    const float t0 = L.s_t[k], t1 = L.s_t[k + 1];
                    const float smid = softplus_march((L.s_sig[k] + L.s_sig[k + 1]) * 0.5f - 1.f);
                    tmid = (t0 + t1) * 0.5f;
                    alpha = 1.f - exp_hw(-(smid * (t1 - t0)));
                }
                const float x = ok ? (1.f - alpha + 1e-10f) : 1.f;
                const float incl = wave_scan_mul(x, lane);
                const float tr = wave_shift_up(incl, 1.f) * carry;
                carry *= wave_last(incl);
                if (ok) { const float wk = alpha * tr; L.w_s[k] = wk; L.trans[k] = tr; acc_w += wk; acc_wt += wk * tmid; }
            }
            w_sum = wave_sum(acc_w);
            wt_sum = wave_sum(acc_wt);
        }
        lds_wave_sync();

        // ---- 2. gradient of the composite.  With q_k = sum_c G[c] colour_k[c]:
        //   dL/dw_k = (q_k + q_{k+1}) / 2 - [white_back] sum_c G[c] + g_depth (tmid_k - depth) / W + g_wsum
        //   dL/dalpha_k = dL/dw_k T_k - (sum_{m>k} dL/dw_m w_m) / (1 - alpha_k + 1e-10)
        const float depth = wt_sum / w_sum;
        const float gd_scale = (w_sum > 0.f && depth == depth) ? g_depth / w_sum : 0.f;     // nan_to_num'd rays pass no depth gradient
        const float gw_const = g_wsum - (p.white_back ? g_sum : 0.f);
        const int n_int = n_all - 1;
        float carry = 0.f;
        for (int kb = 0; kb < n_int; kb += 64) {               // walk the intervals from the far end: suffix sums are prefix sums here
            const int k = n_int - 1 - (kb + lane);
            const bool ok = k >= 0;
            float gw = 0.f, wk = 0.f, t0 = 0.f, t1 = 0.f, smid_in = 0.f;
            if (ok) {
                t0 = L.s_t[k]; t1 = L.s_t[k + 1];
                smid_in = (L.s_sig[k] + L.s_sig[k + 1]) * 0.5f - 1.f;
                wk = L.w_s[k];
                gw = (L.s_q[k] + L.s_q[k + 1]) * 0.5f + gw_const + gd_scale * ((t0 + t1) * 0.5f - depth);
            }
            const float incl = wave_scan_add(gw * wk, lane) + carry;
            carry = wave_last(incl);
            if (ok) {
                const float after = incl - gw * wk;                      // sum over m > k
                const float delta = t1 - t0;
                const float dens = softplus_march(smid_in);
                const float one_minus_alpha = exp_hw(-(dens * delta));
                const float alpha = 1.f - one_minus_alpha;
                const float d_alpha = gw * L.trans[k] - after / (1.f - alpha + 1e-10f);
                const float d_dens = d_alpha * delta * one_minus_alpha;
                const float e = exp_hw(-smid_in);
                const float d_smid = smid_in > 20.f ? d_dens : d_dens * __builtin_amdgcn_rcpf(1.f + e);      // softplus' = sigmoid
                L.ds[k] = d_smid;
            }
        }
        lds_wave_sync();
        for (int q = lane; q < n_all; q += 64) {
            const int e = q < S ? q : fine_e0 + (q - S);
            const int r = L.rank_e[e];
            const float wl = r > 0 ? L.w_s[r - 1] : 0.f, wr = r < n_int ? L.w_s[r] : 0.f;
            const float dl = r > 0 ? L.ds[r - 1] : 0.f, dr = r < n_int ? L.ds[r] : 0.f;
            L.v_e[e] = (wl + wr) * 0.5f;                 // colour of sample r enters intervals r-1 and r with weight 1/2 each
            L.dsig_e[e] = (dl + dr) * 0.5f;              // so does its density
        }
        lds_wave_sync();

        // ---- 3. decoder + lookup gradients
        float* stage_ray = nullptr;
        if constexpr (STAGED) {                         // staged scatter: per ray n_all depths, then n_all rows of 32 floats
            stage_ray = stage + ray * int64_t(n_all) * 33 + n_all;
            for (int k = lane; k < n_all; k += 64) stage[ray * int64_t(n_all) * 33 + k] = L.s_t[k];
        }
        bwd_backward_tiles<STAGED>(P, L, R, grad_planes_item, stage_ray, 0, S, P.tiles_c, G, A, pend, lane);
        if (F > 0) bwd_backward_tiles<STAGED>(P, L, R, grad_planes_item, stage_ray, fine_e0, F, P.tiles_f, G, A, pend, lane);
    }
    if constexpr (!STAGED) {
        if (pend.live > 0) {                           // the last tile's scatter has no next lookup to hide in
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int pl = 0; pl < 3; pl++) bwd_scatter_chunk(L, pend, a, pl, lane);
        }
    }

    bwd_reduce_decoder_grads(A, smem, Gr.grad_w1, Gr.grad_b1, Gr.grad_w2, Gr.grad_b2, tid, lane, j);
}

// ---------------------------------------------------------------------------------------------
// Second kernel of the staged backward on the pipelined path (round 4).  render_kernel_pipe_bwd (render_pipe.inl) has run the ray-level
// part -- forward pipeline, merge, composite gradient -- at the forward kernel's speed and left, per ray and in depth order, the sample
// depths, colour weights v_r and dL/dsigma_r in the staging buffer.  What is left is work per SAMPLE TILE with no ray-level state at all:
// lookup + decoder forward (exact fp32) + the four gradient products + dX rows.  One wave per 16-rank tile, tiles dealt in contiguous
// runs of the locality-ordered ray sequence; the dX rows of a tile are 2 KB of contiguous staging memory (the one-wave-per-ray kernel
// above writes them rank by rank behind a second forward recomputation, with 12 per-sample arrays per wave in LDS).
constexpr int kTileWaves = 4, kTileThreads = 64 * kTileWaves;       // waves of a tile-kernel workgroup
struct DepthListPos {
    const BwdRay& R; const float* dep; float box_scale;
    __device__ __forceinline__ void operator()(int j, float& px, float& py, float& pz) const {
        const float depth = dep[j];
        px = __fadd_rn(R.ox, __fmul_rn(depth, R.dx)) * box_scale;
        py = __fadd_rn(R.oy, __fmul_rn(depth, R.dy)) * box_scale;
        pz = __fadd_rn(R.oz, __fmul_rn(depth, R.dz)) * box_scale;
    }
};

__host__ __device__ inline size_t bwd_tiles_wave_floats() { return 16 * kStagePitch + 16 * kTPitch + 16 * kHPitch + 48; }

// Decoder in LDS for the f16 hi/lo form of the tile kernel: the forward's fragments (stage_decoder<kMlpF16x3>: W1, W2 colour rows as
// hi/lo halves in MFMA fragment order, biases and density row with the base-2 factors folded in), then the two STATIC operands of the
// gradient products in the same hi/lo fragment form, then the density row in true units.
//   g1 [hi|lo][m][lane][8]:      A of dH^T block m = W2c^T:  element e of lane (j, g) = W2[1 + 8g + e][16m + j]
//   g3 [hi|lo][c][s][lane][8]:   A of dX^T block c, k-step s = W1^T: element jj of lane (j, g) = W1[32s + 16(jj>>2) + 4g + (jj&3)][16c + j]
//                                (the k order in which a lane's dPRE registers come: the forward's layer-2 trick)
constexpr int kBwdFragHalves = 4 * 64 * 8;          // one of g1 hi, g1 lo, g3 hi, g3 lo
constexpr int kBwdF16WeightFloats = kWeightFloatsF16 + 64 + 36 + (4 * kBwdFragHalves) / 2 + 64 + 4;
__host__ __device__ constexpr int bwd_tiles_weight_floats() { return kBwdF16WeightFloats > kBwdWeightFloats ? kBwdF16WeightFloats : kBwdWeightFloats; }
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef unsigned u2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ h4 as_h4(unsigned a, unsigned b) { return __builtin_bit_cast(h4, (u2v){a, b}); }
#define GNERF_MFMA16K16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0)

// One 16-sample tile, exact fp32 products (v_mfma_f32_16x16x4_f32): any finite input.
struct BwdTileF32 {
    static constexpr int kMlp = kMlpF32;
    BwdLds L;
    float inv_scale;
    __device__ __forceinline__ void setup(const Params& P, float* smem, const float*, int tid) {
        const gnerf_render_params& p = P.p;
        float* w1 = smem;
        float* w2 = w1 + 64 * kW1Pitch;
        float* b1 = w2 + 33 * kW2Pitch;
        float* b2 = b1 + 64;
        for (int i = tid; i < 64 * 32; i += kTileThreads) w1[(i >> 5) * kW1Pitch + (i & 31)] = p.w1[i];
        for (int i = tid; i < 33 * 64; i += kTileThreads) w2[(i >> 6) * kW2Pitch + (i & 63)] = p.w2[i];
        if (tid < 64) b1[tid] = p.b1[tid];
        if (tid < 36) b2[tid] = tid < 33 ? p.b2[tid] : 0.f;
        __syncthreads();
        L = BwdLds{};
        L.w1 = w1; L.w2 = w2; L.b1 = b1; L.b2 = b2;
        inv_scale = 1.f;
    }
    // X in L.stage, the tile's colour weights in vw, dL/dsigma in dsg, G = dL/d colour sum of channels 16n + j  ->  dX[16][32] in L.tbuf
    __device__ __forceinline__ void tile(const float* vw, const float* dsg, const float (&G)[2], BwdAcc& A, int lane) {
        const int j = lane & 15, g = lane >> 4;
        v4f h[4], o[2];
        float sig;
        bwd_mlp_forward(L, lane, h, o, sig);
        // ---- dO: colour c = 1.002 * s - 0.001 with s = sigmoid(o); dL/dc = G * v_sample  (triplane.py:134, ray_marcher.py:27-45)
        const v4f vs = *reinterpret_cast<const v4f*>(vw + 4 * g);
        const float dsig = dsg[j];
#pragma unroll
        for (int n = 0; n < 2; n++) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const float e = __builtin_amdgcn_exp2f(o[n][r] * -1.44269504088896341f);
                const float s = __builtin_amdgcn_rcpf(1.0f + e);
                const float d = G[n] * vs[r] * (1.002f * s * (1.f - s));
                L.tbuf[(4 * g + r) * kTPitch + 16 * n + j] = d;
                A.b2[n] += d;
            }
        }
        bwd_tile_core(L, h, dsig, A, lane);
    }
    __device__ __forceinline__ void finish(BwdAcc&) {}
};

// The same tile with every product as an error-compensated f16 hi/lo split on the f16 matrix instructions (fp32 accumulation): the
// decoder forward as in the forward kernels (24 v_mfma_f32_16x16x32_f16), dH and dX against static hi/lo fragments (12 + 12), the two
// weight-gradient products -- both operands made at run time, K = the tile's 16 samples -- on v_mfma_f32_16x16x16_f16 (24 + 24): 96
// matrix instructions of 16.5 SIMD cycles where the fp32 form issues 192 of 32.  Loss gradients of 1e-6 are f16-subnormal, and a
// call's gradients span many orders of magnitude (a ray whose weights sum to 1e-6 sends dL/ddepth / 1e-6 down its samples): every
// matrix operand of the gradient side (dO, dsigma, dH, dPRE) is therefore carried MULTIPLIED BY A POWER OF TWO CHOSEN PER TILE from
// the tile's own largest |dO| and |dsigma|, so that the largest |dPRE| possible lands near 2^13.  The scale comes off, exactly, when
// dX is written; the wave's two matrix-accumulated weight gradients (dW1, dW2: 64 registers) are kept in the units of the CURRENT
// tile's scale -- when the scale changes from one tile to the next they are multiplied by the ratio, a power of two (exact; fp32 has
// the range) -- so that the matrix instructions accumulate into them directly.
// Valid under the same guard as the forward's f16 arithmetic (choose_mlp: features, weights, activations in f16's range).
#ifndef GNERF_K2_SPLIT
#define GNERF_K2_SPLIT split_f16x8
#endif
#ifdef GNERF_K2_NO_FENCE
#define GNERF_K2_PHASE_FENCE()
#else
#define GNERF_K2_PHASE_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif
struct BwdTileF16 {
    static constexpr int kMlp = kMlpF16x3;
    CoopLds C;                       // the forward's fragments (w1 = fragment base, w2 = density row * ln2, b1 * log2e, b2 scaled)
    const _Float16* g1; const _Float16* g3;
    const float* ws_true;            // W2[0][:]
    float* stage; float* tbuf; float* hbuf;
    float l1_w2c, max_ws;            // max_hid sum_out |W2c[out][hid]|, max |W2[0]|: |dPRE| <= |dH| <= l1_w2c max|dO| + max_ws max|dsigma|
    float acc_scale;                 // the power of two A.w1 / A.w2 currently carry
    bool sp_direct = false;          // softplus as log2(1 + 2^p') (choose_mlp: every |p'| below exp2's overflow), as the forward kernels take it
    __device__ __forceinline__ void setup(const Params& P, float* smem, const float*, int tid) {
        const gnerf_render_params& p = P.p;
        stage_decoder<kMlpF16x3>(C, smem, p, tid, kTileThreads);
        _Float16* gh = reinterpret_cast<_Float16*>(smem + kWeightFloatsF16 + 64 + 36);
        g1 = gh; g3 = gh + 2 * kBwdFragHalves;
        float* wst = smem + kWeightFloatsF16 + 64 + 36 + (4 * kBwdFragHalves) / 2;
        ws_true = wst;
        for (int i = tid; i < 2048; i += kTileThreads) {
            const int e = i & 7, j = (i >> 3) & 15, g = (i >> 7) & 3, m = i >> 9;             // lane 16 g + j of block m
            const float x = p.w2[(1 + 8 * g + e) * 64 + 16 * m + j];
            const _Float16 hi = (_Float16)x;
            gh[i] = hi;
            gh[kBwdFragHalves + i] = (_Float16)(x - (float)hi);
        }
        for (int i = tid; i < 2048; i += kTileThreads) {
            const int jj = i & 7, j = (i >> 3) & 15, g = (i >> 7) & 3, sx = (i >> 9) & 1, c = i >> 10;     // lane 16 g + j of fragment (c, s)
            const float x = p.w1[(32 * sx + 16 * (jj >> 2) + 4 * g + (jj & 3)) * 32 + 16 * c + j];
            const _Float16 hi = (_Float16)x;
            gh[2 * kBwdFragHalves + i] = hi;
            gh[3 * kBwdFragHalves + i] = (_Float16)(x - (float)hi);
        }
        if (tid < 64) wst[tid] = p.w2[tid];
        __syncthreads();
        const int lane = tid & 63;
        float l1 = 0.f;
        for (int o = 0; o < 32; o++) l1 += fabsf(p.w2[(1 + o) * 64 + lane]);
        float wsm = fabsf(wst[lane]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { l1 = fmaxf(l1, __shfl_xor(l1, o)); wsm = fmaxf(wsm, __shfl_xor(wsm, o)); }
        l1_w2c = l1; max_ws = wsm; acc_scale = 1.f;
    }
    __device__ __forceinline__ void tile(const float* vw, const float* dsg, const float (&G)[2], BwdAcc& A, int lane) {
        const int j = lane & 15, g = lane >> 4;
        const _Float16* w1h = reinterpret_cast<const _Float16*>(C.w1);
        const _Float16* w2h = w1h + 2 * kW1FragHalves;
        // ---- decoder forward (coop_shade_tile's f16 branch): p' = log2(e) pre-activation, hv = log2(1 + 2^p') = H / ln 2, o' = -log2(e) o
        const v4f f_lo = *reinterpret_cast<const v4f*>(stage + j * kStagePitch + 8 * g);
        const v4f f_hi = *reinterpret_cast<const v4f*>(stage + j * kStagePitch + 8 * g + 4);
        const float f[8] = {f_lo[0], f_lo[1], f_lo[2], f_lo[3], f_hi[0], f_hi[1], f_hi[2], f_hi[3]};
        unsigned fh_u[4], fl_u[4];
        GNERF_K2_SPLIT(f, fh_u, fl_u);
        const h8 fh = as_h8((u4v){fh_u[0], fh_u[1], fh_u[2], fh_u[3]}), fl = as_h8((u4v){fl_u[0], fl_u[1], fl_u[2], fl_u[3]});
        v4f hv[4];
        {
            h8 a_hi[4], a_lo[4];
#pragma unroll
            for (int m = 0; m < 4; m++) {
                hv[m] = *reinterpret_cast<const v4f*>(C.b1 + 16 * m + 4 * g);
                a_hi[m] = *reinterpret_cast<const h8*>(w1h + (m * 64 + lane) * 8);
                a_lo[m] = *reinterpret_cast<const h8*>(w1h + kW1FragHalves + (m * 64 + lane) * 8);
            }
#pragma unroll
            for (int m = 0; m < 4; m++) hv[m] = GNERF_MFMA16(a_hi[m], fh, hv[m]);
#pragma unroll
            for (int m = 0; m < 4; m++) hv[m] = GNERF_MFMA16(a_hi[m], fl, hv[m]);
#pragma unroll
            for (int m = 0; m < 4; m++) hv[m] = GNERF_MFMA16(a_lo[m], fh, hv[m]);
        }
#pragma unroll
        for (int m = 0; m < 4; m++) {
            v4f e;
            if (sp_direct) {
#pragma unroll
                for (int r = 0; r < 4; r++) e[r] = __builtin_amdgcn_exp2f(hv[m][r]);
#pragma unroll
                for (int r = 0; r < 4; r++) hv[m][r] = __builtin_amdgcn_logf(1.0f + e[r]);
            } else {
#pragma unroll
                for (int r = 0; r < 4; r++) e[r] = __builtin_amdgcn_exp2f(-fabsf(hv[m][r]));
#pragma unroll
                for (int r = 0; r < 4; r++) e[r] = __builtin_amdgcn_logf(1.0f + e[r]);
#pragma unroll
                for (int r = 0; r < 4; r++) hv[m][r] = fmaxf(hv[m][r] + e[r], e[r]);
            }
            *reinterpret_cast<v4f*>(hbuf + j * kHPitch + 16 * m + 4 * g) = hv[m];               // H / ln2 as [sample][hidden], for dW2's B operand
        }
        GNERF_K2_PHASE_FENCE();
        v4f o[2];
        {
            const float bc0 = C.b2[1 + j], bc1 = C.b2[17 + j];
            o[0] = (v4f){bc0, bc0, bc0, bc0}; o[1] = (v4f){bc1, bc1, bc1, bc1};
            h8 x_hi[2], x_lo[2];
#pragma unroll
            for (int s = 0; s < 2; s++) {
                unsigned xh[4], xl[4];
                const float xs[8] = {hv[2 * s][0], hv[2 * s][1], hv[2 * s][2], hv[2 * s][3], hv[2 * s + 1][0], hv[2 * s + 1][1], hv[2 * s + 1][2], hv[2 * s + 1][3]};
                GNERF_K2_SPLIT(xs, xh, xl);
                x_hi[s] = as_h8((u4v){xh[0], xh[1], xh[2], xh[3]});
                x_lo[s] = as_h8((u4v){xl[0], xl[1], xl[2], xl[3]});
            }
#pragma unroll
            for (int s = 0; s < 2; s++) {
#pragma unroll
                for (int n = 0; n < 2; n++) {
                    const h8 w_hi = *reinterpret_cast<const h8*>(w2h + ((n * 2 + s) * 64 + lane) * 8);
                    const h8 w_lo = *reinterpret_cast<const h8*>(w2h + 2048 + ((n * 2 + s) * 64 + lane) * 8);
                    o[n] = GNERF_MFMA16(x_hi[s], w_hi, o[n]);
                    o[n] = GNERF_MFMA16(x_hi[s], w_lo, o[n]);
                    o[n] = GNERF_MFMA16(x_lo[s], w_hi, o[n]);
                }
            }
        }
        GNERF_K2_PHASE_FENCE();
        // ---- dO: colour c = 1.002 s - 0.001, s = sigmoid(o) = 1 / (1 + 2^o'); kept in registers too (A operand of dW2)
        const v4f vs = *reinterpret_cast<const v4f*>(vw + 4 * g);
        const float dsig_true = dsg[j];
        float dO[8];
        float big = max_ws * fabsf(dsig_true);
#pragma unroll
        for (int n = 0; n < 2; n++) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(o[n][r]));
                const float d = G[n] * vs[r] * (1.002f * s * (1.f - s));
                dO[4 * n + r] = d;
                A.b2[n] += d;
                big = fmaxf(big, l1_w2c * fabsf(d));
            }
        }
        if (g == 0) A.b2s += dsig_true;
        // the tile's scale: |dPRE| <= l1 max|dO| + max_ws max|dsigma| <= 2 max over the lanes of `big`  ->  scaled below 2^13
        // wave maximum of non-negative values on the DPP network (the forward kernels' scan idiom; lane 63 ends up with the maximum)
        big = fmaxf(big, dpp_mov<0x111, 0xf>(0.f, big));
        big = fmaxf(big, dpp_mov<0x112, 0xf>(0.f, big));
        big = fmaxf(big, dpp_mov<0x114, 0xf>(0.f, big));
        big = fmaxf(big, dpp_mov<0x118, 0xf>(0.f, big));
        big = fmaxf(big, dpp_mov<0x142, 0xa>(0.f, big));
        big = fmaxf(big, dpp_mov<0x143, 0xc>(0.f, big));
        big = wave_last(big);
        int ex = 0;
        if (big > 0.f && big < INFINITY) { (void)frexpf(big, &ex); ex = min(max(12 - ex, -60), 60); }
        ex = __builtin_amdgcn_readfirstlane(ex);                    // one value for the wave, whatever the lanes think
        const float scale = ldexpf(1.f, ex), inv = ldexpf(1.f, -ex);
        if (scale != acc_scale) {                                   // wave-uniform: bring the matrix accumulators to this tile's units
            const float ratio = scale * (1.f / acc_scale);          // both powers of two within 2^+-60: exact
#pragma unroll
            for (int m = 0; m < 4; m++) { A.w1[m][0] *= ratio; A.w1[m][1] *= ratio; A.w2[0][m] *= ratio; A.w2[1][m] *= ratio; }
            acc_scale = scale;
        }
        const float dsig = dsig_true * scale;
#pragma unroll
        for (int n = 0; n < 2; n++) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                dO[4 * n + r] *= scale;
                tbuf[(4 * g + r) * kTPitch + 16 * n + j] = dO[4 * n + r];
            }
        }
        lds_wave_sync();
        GNERF_K2_PHASE_FENCE();
        // ---- dH^T[hidden][sample] = W2c^T dO^T (+ the density row on the vector ALU): B = this lane's sample row of dO, outputs 8g..8g+7
        v4f dh[4];
        {
            const v4f b_lo4 = *reinterpret_cast<const v4f*>(tbuf + j * kTPitch + 8 * g);
            const v4f b_hi4 = *reinterpret_cast<const v4f*>(tbuf + j * kTPitch + 8 * g + 4);
            const float bv[8] = {b_lo4[0], b_lo4[1], b_lo4[2], b_lo4[3], b_hi4[0], b_hi4[1], b_hi4[2], b_hi4[3]};
            unsigned bh_u[4], bl_u[4];
            GNERF_K2_SPLIT(bv, bh_u, bl_u);
            const h8 bh = as_h8((u4v){bh_u[0], bh_u[1], bh_u[2], bh_u[3]}), bl = as_h8((u4v){bl_u[0], bl_u[1], bl_u[2], bl_u[3]});
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const v4f ws = *reinterpret_cast<const v4f*>(ws_true + 16 * m + 4 * g);
                dh[m] = ws * dsig;
                A.w2s[m] += hv[m] * dsig_true;                                                   // (x ln2 at the end)
                const h8 a_hi = *reinterpret_cast<const h8*>(g1 + (m * 64 + lane) * 8);
                const h8 a_lo = *reinterpret_cast<const h8*>(g1 + kBwdFragHalves + (m * 64 + lane) * 8);
                dh[m] = GNERF_MFMA16(a_hi, bh, dh[m]);
                dh[m] = GNERF_MFMA16(a_hi, bl, dh[m]);
                dh[m] = GNERF_MFMA16(a_lo, bh, dh[m]);
            }
        }
        GNERF_K2_PHASE_FENCE();
        // ---- dW2c[out][hidden] += dO^T (H / ln2): A = the dO values this lane computed (samples 4g..4g+3 of outputs 16o + j), B from hbuf
        {
            unsigned ah_u[4], al_u[4];
            GNERF_K2_SPLIT(dO, ah_u, al_u);
#pragma unroll
            for (int np = 0; np < 2; np++) {                       // hidden blocks 2 np, 2 np + 1
                float bvals[8];
#pragma unroll
                for (int q = 0; q < 2; q++)
#pragma unroll
                    for (int e = 0; e < 4; e++) bvals[4 * q + e] = hbuf[(4 * g + e) * kHPitch + 16 * (2 * np + q) + j];
                unsigned bh_u[4], bl_u[4];
                GNERF_K2_SPLIT(bvals, bh_u, bl_u);
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    const h4 b_hi = as_h4(bh_u[2 * q], bh_u[2 * q + 1]), b_lo = as_h4(bl_u[2 * q], bl_u[2 * q + 1]);
#pragma unroll
                    for (int oo = 0; oo < 2; oo++) {
                        const h4 a_hi = as_h4(ah_u[2 * oo], ah_u[2 * oo + 1]), a_lo = as_h4(al_u[2 * oo], al_u[2 * oo + 1]);
                        v4f acc = A.w2[oo][2 * np + q];
                        acc = GNERF_MFMA16K16(a_hi, b_hi, acc);
                        acc = GNERF_MFMA16K16(a_hi, b_lo, acc);
                        acc = GNERF_MFMA16K16(a_lo, b_hi, acc);
                        A.w2[oo][2 * np + q] = acc;
                    }
                }
            }
        }
        GNERF_K2_PHASE_FENCE();
        // ---- through softplus: d/dpre softplus(pre) = 1 - exp(-H) = 1 - 2^-(H / ln2)
#pragma unroll
        for (int m = 0; m < 4; m++) {
#pragma unroll
            for (int r = 0; r < 4; r++) dh[m][r] *= 1.f - __builtin_amdgcn_exp2f(-hv[m][r]);
            A.b1[m] += dh[m] * inv;
        }
        lds_wave_sync();                                            // every lane has read H from hbuf and its dO row from tbuf
        GNERF_K2_PHASE_FENCE();
        // ---- dX^T[channel][sample] = W1^T dPRE^T: B straight from this lane's dPRE registers (k order of the g3 fragments)
        v4f dx[2];
        dx[0] = dx[1] = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 2; s++) {
            const float pv[8] = {dh[2 * s][0], dh[2 * s][1], dh[2 * s][2], dh[2 * s][3], dh[2 * s + 1][0], dh[2 * s + 1][1], dh[2 * s + 1][2], dh[2 * s + 1][3]};
            unsigned ph_u[4], pl_u[4];
            GNERF_K2_SPLIT(pv, ph_u, pl_u);
            const h8 p_hi = as_h8((u4v){ph_u[0], ph_u[1], ph_u[2], ph_u[3]}), p_lo = as_h8((u4v){pl_u[0], pl_u[1], pl_u[2], pl_u[3]});
#pragma unroll
            for (int c = 0; c < 2; c++) {
                const h8 a_hi = *reinterpret_cast<const h8*>(g3 + ((c * 2 + s) * 64 + lane) * 8);
                const h8 a_lo = *reinterpret_cast<const h8*>(g3 + kBwdFragHalves + ((c * 2 + s) * 64 + lane) * 8);
                dx[c] = GNERF_MFMA16(a_hi, p_hi, dx[c]);
                dx[c] = GNERF_MFMA16(a_hi, p_lo, dx[c]);
                dx[c] = GNERF_MFMA16(a_lo, p_hi, dx[c]);
            }
        }
#pragma unroll
        for (int m = 0; m < 4; m++) *reinterpret_cast<v4f*>(hbuf + j * kHPitch + 16 * m + 4 * g) = dh[m];         // dPRE[sample][hidden]
        *reinterpret_cast<v4f*>(tbuf + j * kTPitch + 4 * g) = dx[0] * inv;                                         // dX[sample][channel], true units
        *reinterpret_cast<v4f*>(tbuf + j * kTPitch + 16 + 4 * g) = dx[1] * inv;
        lds_wave_sync();
        GNERF_K2_PHASE_FENCE();
        // ---- dW1[hidden][channel] += dPRE^T X: A = dPRE^T from hbuf, B = X^T from the staged features
        {
            float xv[8];
#pragma unroll
            for (int c = 0; c < 2; c++)
#pragma unroll
                for (int e = 0; e < 4; e++) xv[4 * c + e] = stage[(4 * g + e) * kStagePitch + 16 * c + j];
            unsigned xh_u[4], xl_u[4];
            GNERF_K2_SPLIT(xv, xh_u, xl_u);
#pragma unroll
            for (int mp = 0; mp < 2; mp++) {
                float av[8];
#pragma unroll
                for (int q = 0; q < 2; q++)
#pragma unroll
                    for (int e = 0; e < 4; e++) av[4 * q + e] = hbuf[(4 * g + e) * kHPitch + 16 * (2 * mp + q) + j];
                unsigned ah_u[4], al_u[4];
                GNERF_K2_SPLIT(av, ah_u, al_u);
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    const h4 a_hi = as_h4(ah_u[2 * q], ah_u[2 * q + 1]), a_lo = as_h4(al_u[2 * q], al_u[2 * q + 1]);
#pragma unroll
                    for (int c = 0; c < 2; c++) {
                        const h4 b_hi = as_h4(xh_u[2 * c], xh_u[2 * c + 1]), b_lo = as_h4(xl_u[2 * c], xl_u[2 * c + 1]);
                        v4f acc = A.w1[2 * mp + q][c];
                        acc = GNERF_MFMA16K16(a_hi, b_hi, acc);
                        acc = GNERF_MFMA16K16(a_hi, b_lo, acc);
                        acc = GNERF_MFMA16K16(a_lo, b_hi, acc);
                        A.w1[2 * mp + q][c] = acc;
                    }
                }
            }
        }
    }
    // the matrix accumulators carry acc_scale, and dW2 was accumulated against H / ln2
    __device__ __forceinline__ void finish(BwdAcc& A) {
        const float k1 = 1.f / acc_scale, k2 = k1 * kLn2;
#pragma unroll
        for (int m = 0; m < 4; m++) { A.w1[m][0] *= k1; A.w1[m][1] *= k1; A.w2[0][m] *= k2; A.w2[1][m] *= k2; A.w2s[m] *= kLn2; }
    }
};

template <class Tile>
__device__ __forceinline__ void render_bwd_tiles_body(const Params& P, const gnerf_render_grads& Gr, float* stage, float* smem, bool sp_direct = false) {
    const gnerf_render_params& p = P.p;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n_all = p.depth_resolution + p.depth_resolution_importance;
    const float* tail = stage + int64_t(P.total_rays) * P.bwd_ray_stride;
    Tile K;
    K.setup(P, smem, tail, tid);
    if constexpr (Tile::kMlp != kMlpF32) K.sp_direct = sp_direct;
    float* base = smem + bwd_tiles_weight_floats() + size_t(wv) * bwd_tiles_wave_floats();
    BwdLds L = {};
    L.stage = base; L.tbuf = L.stage + 16 * kStagePitch; L.hbuf = L.tbuf + 16 * kTPitch;
    if constexpr (Tile::kMlp == kMlpF32) { K.L.stage = L.stage; K.L.tbuf = L.tbuf; K.L.hbuf = L.hbuf; }
    else { K.stage = L.stage; K.tbuf = L.tbuf; K.hbuf = L.hbuf; }
    float* dep = L.hbuf + 16 * kHPitch;            // [16] depths, [16] colour weights, [16] dL/dsigma of the tile
    float* vw = dep + 16;
    float* dsg = vw + 16;
    BwdAcc A;
#pragma unroll
    for (int m = 0; m < 4; m++) A.w1[m][0] = A.w1[m][1] = A.w2[0][m] = A.w2[1][m] = A.w2s[m] = A.b1[m] = (v4f){0.f, 0.f, 0.f, 0.f};
    A.b2[0] = A.b2[1] = A.b2s = 0.f;
    const int j = lane & 15;
    const int64_t plane_floats = int64_t(3) * p.plane_h * p.plane_w * 32;
    // tiles of the locality-ordered ray sequence (pipe_seq_to_ray): XCD x owns a contiguous eighth (workgroups b, b+8, ... share an
    // XCD), cut into equal contiguous runs for the XCD's waves
    const int tiles_per_ray = (n_all + 15) >> 4;
    const int64_t total_seq = (P.tiles_per_item > 0 || linear_pad(P) > 0) ? int64_t(P.n_tiles) * 16 : int64_t(P.total_rays);
    const int xcd = blockIdx.x % kNumXCD, wg = blockIdx.x / kNumXCD, wgs = gridDim.x / kNumXCD;
    const int64_t x0 = total_seq * xcd / kNumXCD * tiles_per_ray, x1 = total_seq * (xcd + 1) / kNumXCD * tiles_per_ray;
    const int64_t waves = int64_t(wgs) * kTileWaves, me = int64_t(wg) * kTileWaves + wv;
    const int64_t t0 = x0 + (x1 - x0) * me / waves, t1 = x0 + (x1 - x0) * (me + 1) / waves;
    for (int64_t t = t0; t < t1; t++) {
        const int64_t seq = t / tiles_per_ray;
        const int T = int(t - seq * tiles_per_ray);
        const int ray = pipe_seq_to_ray<true>(P, seq);
        if (ray < 0) continue;
        const int item = ray / p.rays_per_item;
        const int live = min(16, n_all - 16 * T);
        float* const ray_block = stage + int64_t(ray) * P.bwd_ray_stride;
        float* const rows = ray_block + n_all + T * P.bwd_tile_pitch;
        if (lane < 16) dep[lane] = ray_block[16 * T + min(lane, live - 1)];
        else if (lane < 48) dep[lane] = ((lane & 15) < live) ? rows[lane - 16] : 0.f;       // vw[0..15], dsg[0..15]
        BwdRay R;
        R.ox = p.ray_origins[int64_t(ray) * 3 + 0]; R.oy = p.ray_origins[int64_t(ray) * 3 + 1]; R.oz = p.ray_origins[int64_t(ray) * 3 + 2];
        R.dx = p.ray_dirs[int64_t(ray) * 3 + 0]; R.dy = p.ray_dirs[int64_t(ray) * 3 + 1]; R.dz = p.ray_dirs[int64_t(ray) * 3 + 2];
        R.planes = reinterpret_cast<const char*>(p.planes_nhwc + int64_t(item) * plane_floats);
        float G[2];
        G[0] = Gr.grad_rgb ? 2.f * Gr.grad_rgb[int64_t(ray) * 32 + j] : 0.f;
        G[1] = Gr.grad_rgb ? 2.f * Gr.grad_rgb[int64_t(ray) * 32 + 16 + j] : 0.f;
        lds_wave_sync();
        bwd_gather_tile_rolling(P, L, R.planes, DepthListPos{R, dep, P.box_scale}, lane);
        K.tile(vw, dsg, G, A, lane);
        if (Gr.grad_planes_nhwc) {                                  // the tile's dX rows: 16 x 128 contiguous bytes
            const int half = lane >> 5, ch = lane & 31;
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const int smp = 2 * q + half;
                if (smp < live) rows[smp * 32 + ch] = L.tbuf[smp * kTPitch + ch];
            }
        }
        lds_wave_sync();
    }
    K.finish(A);
    bwd_reduce_decoder_grads(A, smem, Gr.grad_w1, Gr.grad_b1, Gr.grad_w2, Gr.grad_b2, tid, lane, j, kTileThreads);
}

// Decoder arithmetic as in the forward: the f16 hi/lo form (1.45 ms at config 2; plane and decoder gradients within 1e-6 of the fp32
// form) when the device-side range check allows it, the exact-fp32 form (2.12 ms) otherwise.
// Until the middle of round 4 the f16 form was opt-in: with two workgroups per CU -- the launch shape that gives the speed -- a
// handful of samples per launch (of 2e5) came out wrong, different ones from run to run.  Root cause (tools/dbg_bwd_stage.py, assembly
// variants of this kernel; profiles/r04_pk_opsel_hazard.md): ONE instruction of the dO block, `v_pk_mul_f32 D, S0, S1 op_sel:[0,1]`
// (low half of the result from the HIGH register of src1), reads that register as 0.0 in lanes 48-63 now and then while the other
// wave of its SIMD has a v_mfma_f32_16x16x32_f16 in flight: sample 13 of a tile lost its first 16 dO entries.  The product with
// the operands exchanged (`op_sel:[1,0]`) is exact in every run; the build (csrc/compile_unit.sh, pk_opsel_fixup.py) rewrites every
// such instruction of the library and tools/isa_lint.py checks the result.
__global__ __launch_bounds__(kTileThreads, 2) void render_bwd_tiles_kernel(Params P, gnerf_render_grads Gr, float* stage) {
    extern __shared__ __align__(16) float smem[];
    int mlp = P.p.mlp_mode;
    bool sp_direct = false;
    if (mlp == kMlpAuto) mlp = choose_mlp(P, smem, &sp_direct);
    if (mlp == kMlpF32) render_bwd_tiles_body<BwdTileF32>(P, Gr, stage, smem);
    else                render_bwd_tiles_body<BwdTileF16>(P, Gr, stage, smem, sp_direct);
}

// ---------------------------------------------------------------------------------------------
// Staged plane-gradient scatter, second pass (round 2).  render_bwd_kernel above is bound by the device's float-atomic rate: every
// (sample, tap) costs two 64-byte atomic requests that are resolved at the memory side, 149 M per launch at BASELINE config 2.  A
// 4x4-pixel ray tile touches each texel ~4-5 times, mostly from samples at similar depth on neighbouring rays, but the first
// pass has neither LDS (2 x 79 KB per CU) nor registers (256) left to aggregate them.  So, when the caller provides a staging
// buffer, the first pass writes each sample's dX row in its ray's merged DEPTH ORDER and this kernel does the scatter: one
// workgroup per 16-ray tile walks the tile's samples in chunks of kScatterRanks depth ranks (all 16 rays at once) and SORTS the
// chunk's 2 304 tap contributions by texel with a counting sort in LDS -- texel -> table slot by compare-and-swap on integer tags
// (open addressing at a load of ~0.3), a counter per slot that also hands each contribution its place in the slot's bucket, a prefix
// sum, a scatter of (sample, weight) pairs -- after which every half-wave sums whole buckets from the chunk's dX rows in LDS and
// sends ONE global atomic per texel and channel.  (A first version accumulated into an LDS table with float LDS atomics: those
// run at ~0.4 lanes per cycle here and made this pass 14 ms; the sort needs integer atomics only, 2 304 per chunk.)
// A contribution that finds no free slot goes straight to global memory, so correctness never depends on the table.
constexpr int kScatterThreads = 1024, kScatterSlots = 2048, kScatterRanks = 16, kScatterProbes = 16;
constexpr unsigned kScatterEmpty = 0xffffffffu;
constexpr int kScatterSamples = 16 * kScatterRanks;             // samples of one chunk
constexpr int kScatterEntries = kScatterSamples * 12;           // tap contributions of one chunk
static_assert(kScatterSlots <= (1 << 11) && kScatterSamples * 32 <= (1 << 13), "an entry packs slot and dX row offset into 24 bits");
static_assert(kScatterSamples * 3 <= kScatterThreads && kScatterSlots == 2 * kScatterThreads, "one thread per (sample, plane), two slots per thread in the scan");
__host__ __device__ inline size_t scatter_lds_floats() {
    return 16 * 8 + size_t(kScatterSamples) * 32 + 2 * size_t(kScatterSlots) + 2 + 2 * size_t(kScatterEntries) + 32;
}

__global__ __launch_bounds__(kScatterThreads, 8) void plane_scatter_kernel(Params P, const float* __restrict__ stage, float* __restrict__ grad_planes) {
    extern __shared__ __align__(16) float smem[];
    const gnerf_render_params& p = P.p;
    float* rays = smem;                                                        // [16][o, d, ray id, -]
    float* dx = rays + 16 * 8;                                                 // [samples][32]
    unsigned* tag = reinterpret_cast<unsigned*>(dx + kScatterSamples * 32);    // [slots] texel key, or empty
    int* cnt = reinterpret_cast<int*>(tag + kScatterSlots);                    // [slots + 1] bucket sizes, then (in place) bucket starts
    float2* entry = reinterpret_cast<float2*>(cnt + kScatterSlots + 2);        // [entries] (sample of the chunk, bilinear weight)
    int* wave_tot = reinterpret_cast<int*>(entry + kScatterEntries);           // [16]
    const int tid = threadIdx.x, lane = tid & 63, ch = lane & 31, wv = tid >> 6;
    const int hw = tid >> 5;                                                   // half-wave index
    const int n_all = p.depth_resolution + p.depth_resolution_importance;
    const int tile = blockIdx.x;
    const int H = p.plane_h, W = p.plane_w;
    if (tid < 16) {                                                            // the tile's 16 rays (the first pass's tile order)
        int64_t ray = -1;
        if (P.tiles_per_item > 0) {
            const int item = tile / P.tiles_per_item, tt = tile % P.tiles_per_item;
            const int tx = tt / P.tiles_y, ty = tt % P.tiles_y;
            ray = int64_t(item) * p.rays_per_item + int64_t(ty * 4 + (tid >> 2)) * p.image_width + tx * 4 + (tid & 3);
        } else {
            ray = int64_t(tile) * 16 + tid;
            if (ray >= P.total_rays) ray = -1;
        }
        float* r = rays + tid * 8;
        if (ray >= 0) {
            r[0] = p.ray_origins[ray * 3 + 0]; r[1] = p.ray_origins[ray * 3 + 1]; r[2] = p.ray_origins[ray * 3 + 2];
            r[3] = p.ray_dirs[ray * 3 + 0];    r[4] = p.ray_dirs[ray * 3 + 1];    r[5] = p.ray_dirs[ray * 3 + 2];
        }
        reinterpret_cast<int*>(r)[6] = int(ray);
    }
    for (int i = tid; i < kScatterSlots; i += kScatterThreads) { tag[i] = kScatterEmpty; cnt[i] = 0; }
    __syncthreads();
    const int first_ray = reinterpret_cast<const int*>(rays)[6];
    if (first_ray < 0) return;                                                 // (uniform: tiles are filled from ray 0 of the tile)
    const int item = first_ray / p.rays_per_item;                             // a tile never straddles items (checked by the launcher)
    float* grad_item = grad_planes + int64_t(item) * 3 * H * W * 32;

    // A chunk's staged values (its dX rows: eight per half-wave; the depth of this thread's sample) are fetched into registers one chunk
    // AHEAD, at the top of the previous chunk, and made to land before that chunk's global atomics go out.  Loads, stores and
    // returnless atomics share one completion counter (vmcnt) and complete out of order with respect to each other, so a wait for a
    // load behind atomics is a wait for the atomics' acknowledgements too: fetching a chunk at its own top put every chunk behind the
    // round trips of the previous chunk's atomics.
    constexpr int kRowsPerHw = kScatterSamples / (kScatterThreads / 32);
    static_assert(kRowsPerHw * (kScatterThreads / 32) == kScatterSamples, "dX rows divide over the half-waves");
    float pre_dx[kRowsPerHw], pre_depth = 0.f;
    auto prefetch = [&](int k0) {
        const int nk = min(kScatterRanks, n_all - k0);
        const int n_smp = 16 * nk;
#pragma unroll
        for (int q = 0; q < kRowsPerHw; q++) {
            const int sr = hw + q * (kScatterThreads / 32);
            pre_dx[q] = 0.f;
            if (sr < n_smp) {
                const int ray = reinterpret_cast<const int*>(rays + (sr / nk) * 8)[6];
                if (ray >= 0) pre_dx[q] = stage[int64_t(ray) * n_all * 33 + n_all + int64_t(k0 + sr % nk) * 32 + ch];
            }
        }
        const int sr3 = tid / 3;
        pre_depth = 0.f;
        if (sr3 < n_smp) {
            const int ray = reinterpret_cast<const int*>(rays + (sr3 / nk) * 8)[6];
            if (ray >= 0) pre_depth = stage[int64_t(ray) * n_all * 33 + k0 + sr3 % nk];
        }
    };
    prefetch(0);

    for (int k0 = 0; k0 < n_all; k0 += kScatterRanks) {
        const int nk = min(kScatterRanks, n_all - k0);
        const int n_smp = 16 * nk;
        // ---- the chunk's dX rows into LDS (sample sr = ray-in-tile * nk + rank-in-chunk)
#pragma unroll
        for (int q = 0; q < kRowsPerHw; q++) {
            const int sr = hw + q * (kScatterThreads / 32);
            if (sr < n_smp) dx[sr * 32 + ch] = pre_dx[q];
        }
        const float my_depth = pre_depth;
        if (k0 + kScatterRanks < n_all) prefetch(k0 + kScatterRanks);         // in flight during the table / prefix / bucket phases below
        // ---- one thread per (sample, plane): its four taps, a table slot and a place in the slot's bucket for each
        int sp[4] = {-1, -1, -1, -1};                                        // table slot (low 12 bits) | place in its bucket << 12; -1 = no slot
        unsigned keys[4] = {0, 0, 0, 0};
        v4f wgt = {0.f, 0.f, 0.f, 0.f};
        const int my_sr = tid / 3, my_pl = tid % 3;
        const bool mine = my_sr < n_smp;
        if (mine) {
            const float* r = rays + (my_sr / nk) * 8;
            const int ray = reinterpret_cast<const int*>(r)[6];
            if (ray >= 0) {
                const float depth = my_depth;
                const float px = __fadd_rn(r[0], __fmul_rn(depth, r[3])) * P.box_scale;
                const float py = __fadd_rn(r[1], __fmul_rn(depth, r[4])) * P.box_scale;
                const float pz = __fadd_rn(r[2], __fmul_rn(depth, r[5])) * P.box_scale;
                const float u = my_pl == 2 ? pz : px;
                const float v = my_pl == 0 ? py : (my_pl == 1 ? pz : px);
                uint4 off;
                plane_taps(H, W, u, v, P.tex_pitch, P.row_pitch, unsigned(my_pl) * P.plane_pitch, off, wgt);
                keys[0] = off.x; keys[1] = off.y; keys[2] = off.z; keys[3] = off.w;
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    if (wgt[t] != 0.f) {
                        const unsigned h = ((keys[t] >> 7) * 2654435761u) >> 21;               // 11 bits
                        for (int pr = 0; pr < kScatterProbes; pr++) {
                            const unsigned s2 = (h + pr) & (kScatterSlots - 1);
                            const unsigned old = atomicCAS(tag + s2, kScatterEmpty, keys[t]);
                            if (old == kScatterEmpty || old == keys[t]) { sp[t] = int(s2); break; }
                        }
                        if (sp[t] >= 0) sp[t] |= atomicAdd(cnt + sp[t], 1) << 12;
                    }
                }
            }
        }
        __syncthreads();
        // ---- exclusive prefix sum of the bucket sizes (two adjacent slots per thread)
        {
            const int c0 = cnt[2 * tid], c1 = cnt[2 * tid + 1];
            int incl = c0 + c1;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int up = __shfl_up(incl, o); if (lane >= o) incl += up; }
            if (lane == 63) wave_tot[wv] = incl;
            __syncthreads();
            int base = 0;
            for (int w2 = 0; w2 < wv; w2++) base += wave_tot[w2];
            cnt[2 * tid] = base + incl - c0 - c1;                              // (every size was read before the barrier above)
            cnt[2 * tid + 1] = base + incl - c1;
            if (tid == kScatterThreads - 1) cnt[kScatterSlots] = base + incl;
        }
        __syncthreads();
        // ---- contributions to their buckets; the rare ones without a slot go straight to memory (32 channels from one lane)
        if (mine) {
#pragma unroll
            for (int t = 0; t < 4; t++) {
                if (wgt[t] != 0.f) {
                    if (sp[t] >= 0) {
                        entry[cnt[sp[t] & 4095] + (sp[t] >> 12)] = make_float2(__int_as_float(((sp[t] & 4095) << 13) | (my_sr * 32)), wgt[t]);
                    } else {
                        for (int c2 = 0; c2 < 32; c2++) {
                            const float c = dx[my_sr * 32 + c2] * wgt[t];
                            if (c != 0.f) unsafeAtomicAdd(grad_item + (keys[t] >> 2) + c2, c);
                        }
                    }
                }
            }
        }
        __syncthreads();
        // ---- every half-wave walks an equal share of the SORTED contributions (lane = channel), sums runs of one texel from the dX
        // rows in LDS and sends one global atomic per run and channel (a texel whose bucket straddles two shares gets two).
        // Walking entries, four per LDS round trip, instead of the table's slots (two dependent reads per slot, most of them
        // empty) took this pass from 3.16 to 2.70 ms at config 2 (its atomic requests alone need 2.15 ms).
        {
            // the next chunk's values have to be in their registers NOW, before this chunk's atomics are in flight (see above); an
            // empty asm that "rewrites" them makes the compiler wait here and nowhere later
#pragma unroll
            for (int q = 0; q < kRowsPerHw; q++) asm volatile("" : "+v"(pre_dx[q]));
            asm volatile("" : "+v"(pre_depth));
            const int total = cnt[kScatterSlots];
            const int share = (total + kScatterThreads / 32 - 1) / (kScatterThreads / 32);
            const int e0 = hw * share, e1 = min(total, e0 + share);
            int cur = -1;
            float sum = 0.f;
            for (int i = e0; i < e1; i += 4) {
                float2 e[4];
                float v[4];
#pragma unroll
                for (int u = 0; u < 4; u++) e[u] = entry[min(i + u, e1 - 1)];
#pragma unroll
                for (int u = 0; u < 4; u++) v[u] = dx[(__float_as_int(e[u].x) & 8191) + ch];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    if (i + u < e1) {
                        const int sl = __float_as_int(e[u].x) >> 13;
                        if (sl != cur) {
                            if (cur >= 0 && sum != 0.f) unsafeAtomicAdd(grad_item + (tag[cur] >> 2) + ch, sum);
                            cur = sl;
                            sum = 0.f;
                        }
                        sum = fmaf(e[u].y, v[u], sum);
                    }
                }
            }
            if (cur >= 0 && sum != 0.f) unsafeAtomicAdd(grad_item + (tag[cur] >> 2) + ch, sum);
        }
        __syncthreads();
        for (int i = tid; i < kScatterSlots; i += kScatterThreads) { tag[i] = kScatterEmpty; cnt[i] = 0; }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// Backward of run_model for arbitrary points (renderer.py:142-148; TriPlaneGenerator.sample_mixed, which the density
// regulariser of training differentiates): the gradient arrives per point -- dL/dsigma [n_items, n_points, 1] and
// dL/drgb [n_items, n_points, 32] -- so there is no composite to undo; each wave takes 16-point tiles through the same
// lookup + MLP forward + decoder backward + deferred plane scatter as the renderer's backward.
struct QueryBwdArgs {
    const float* points; const float* grad_sigma; const float* grad_rgb;      // any of the two gradients may be NULL
    int n_points, tiles_per_item, n_tiles;
    float* grad_planes_nhwc; float* grad_w1; float* grad_b1; float* grad_w2; float* grad_b2;
};

struct PointTilePos {
    const float* pts; int n_points, tile; float box_scale;
    __device__ __forceinline__ void operator()(int j, float& px, float& py, float& pz) const {
        const int idx = min(16 * tile + j, n_points - 1);
        px = pts[idx * 3 + 0] * box_scale; py = pts[idx * 3 + 1] * box_scale; pz = pts[idx * 3 + 2] * box_scale;
    }
};

__global__ __launch_bounds__(kBwdThreads, 2) void query_bwd_kernel(Params P, QueryBwdArgs Q) {
    extern __shared__ __align__(16) float smem[];
    const gnerf_render_params& p = P.p;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    float* w1 = smem;
    float* w2 = w1 + 64 * kW1Pitch;
    float* b1 = w2 + 33 * kW2Pitch;
    float* b2 = b1 + 64;
    for (int i = tid; i < 64 * 32; i += kBwdThreads) w1[(i >> 5) * kW1Pitch + (i & 31)] = p.w1[i];
    for (int i = tid; i < 33 * 64; i += kBwdThreads) w2[(i >> 6) * kW2Pitch + (i & 63)] = p.w2[i];
    if (tid < 64) b1[tid] = p.b1[tid];
    if (tid < 36) b2[tid] = tid < 33 ? p.b2[tid] : 0.f;
    __syncthreads();
    BwdLds L = {};
    L.w1 = w1; L.w2 = w2; L.b1 = b1; L.b2 = b2;
    float* base = smem + kBwdWeightFloats + size_t(wv) * bwd_wave_floats(0);
    L.stage = base; L.tbuf = L.stage + 16 * kStagePitch; L.hbuf = L.tbuf + 16 * kTPitch; L.taps = L.hbuf + 16 * kHPitch;
    BwdAcc A;
#pragma unroll
    for (int m = 0; m < 4; m++) A.w1[m][0] = A.w1[m][1] = A.w2[0][m] = A.w2[1][m] = A.w2s[m] = A.b1[m] = (v4f){0.f, 0.f, 0.f, 0.f};
    A.b2[0] = A.b2[1] = A.b2s = 0.f;
    BwdPending pend = {nullptr, 0};
    const int j = lane & 15, g = lane >> 4;
    const int64_t plane_floats = int64_t(3) * p.plane_h * p.plane_w * 32;
    for (int tile = blockIdx.x * kBwdWaves + wv; tile < Q.n_tiles; tile += gridDim.x * kBwdWaves) {
        const int item = tile / Q.tiles_per_item, t = tile % Q.tiles_per_item;
        const char* planes = reinterpret_cast<const char*>(p.planes_nhwc + int64_t(item) * plane_floats);
        const float* pts = Q.points + int64_t(item) * Q.n_points * 3;
        bwd_gather_tile<true>(P, L, planes, PointTilePos{pts, Q.n_points, t, P.box_scale}, pend, lane);
        v4f h[4], o[2];
        float sig;
        bwd_mlp_forward(L, lane, h, o, sig);
        const int64_t pt0 = int64_t(item) * Q.n_points + 16 * t;
        const float dsig = (Q.grad_sigma && 16 * t + j < Q.n_points) ? Q.grad_sigma[pt0 + j] : 0.f;
#pragma unroll
        for (int n = 0; n < 2; n++) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int q = 4 * g + r;
                const float gc = (Q.grad_rgb && 16 * t + q < Q.n_points) ? Q.grad_rgb[(pt0 + q) * 32 + 16 * n + j] : 0.f;
                const float e = __builtin_amdgcn_exp2f(o[n][r] * -1.44269504088896341f);
                const float s = __builtin_amdgcn_rcpf(1.0f + e);
                const float d = gc * (1.002f * s * (1.f - s));                  // rgb = 1.002 sigmoid(o) - 0.001 (triplane.py:134)
                L.tbuf[q * kTPitch + 16 * n + j] = d;
                A.b2[n] += d;
            }
        }
        bwd_tile_core(L, h, dsig, A, lane);
        if (Q.grad_planes_nhwc) { pend.base = Q.grad_planes_nhwc + int64_t(item) * plane_floats; pend.live = min(16, Q.n_points - 16 * t); }
        lds_wave_sync();
    }
    if (pend.live > 0) {
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int pl = 0; pl < 3; pl++) bwd_scatter_chunk(L, pend, a, pl, lane);
    }
    bwd_reduce_decoder_grads(A, smem, Q.grad_w1, Q.grad_b1, Q.grad_w2, Q.grad_b2, tid, lane, j);
}
