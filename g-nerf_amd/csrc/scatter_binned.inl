// Plane-gradient scatter WITHOUT global float atomics (round 5).  Included by render.hip after render_bwd.inl.
//
// The reference's plane gradient is F.grid_sample's backward (renderer.py:55-65 through autograd; grid_sample_gradfix.py:57-77 is the
// same operator): one float atomic per tap and channel.  plane_scatter_kernel (render_bwd.inl) cut the atomics 3.8x by merging a ray
// tile's taps in LDS, and then sat at 0.79 of the device's float-atomic rate -- 1.33 TB/s of 64-byte requests, all resolved at the
// memory side (profiles/r04_atomic_scope_probe.txt): the approach, not the kernel, was the ceiling.  Here the staged dX rows are
// BINNED by plane tile and each tile's sum is formed in LDS by the one workgroup that owns the tile, then written with plain stores:
//
//   bin_count      one workgroup per 16-ray tile: every (sample, plane) whose footprint has a non-zero tap is a RECORD of the plane tile
//                  (16 x 16 texels) its top-left tap falls in; per tile: record count and max |dX| over its records
//   bin_scan       exclusive scan of the counts (rounded up to whole batches of four records) -> each tile's segment of the record
//                  arrays; the fixed-point scale of every tile; work order (largest tiles first)
//   bin_fill       the same walk again: per record the float index of its dX row, its cell in the tile, and its four tap weights
//                  already multiplied by the tile's scale
//   bin_accumulate persistent workgroups pull tiles: 17 x 17 texels x 32 channels of 64-bit FIXED-POINT accumulators in LDS (the tile
//                  plus the one-texel halo a footprint can reach to the right / below).  A WAVE takes a record at a time: the record
//                  comes through the scalar cache (s_load: row, cell and weights are wave-uniform, no vector instruction touches
//                  them), the 32 lanes of each half read the record's 128-byte dX row, half h adds the two taps of footprint row h with
//                  ds_add_u64.  The tile's own 16 x 16 texels are then converted to fp32 and ADDED to grad_planes with plain loads /
//                  stores, the halo goes to a side buffer
//   bin_halo       every tile adds the halos of its left, upper and upper-left neighbours (fixed order) to its first column / row
//
// Fixed point: a contribution c = w * dx is added as floor(c * 2^s) with ONE scale per tile, s = 61 - ceil(log2(n_t)) - exponent(B_t)
// from the tile's record count n_t and bound B_t >= |dx| (so that n_t contributions cannot overflow 63 bits).  Integer addition is
// associative: the sum does not depend on the order in which records arrive, i.e. plane gradients are BIT-REPRODUCIBLE from run to
// run -- which neither the float-atomic forms nor the reference's grid_sampler_2d_backward are -- and each texel is rounded to fp32
// once, from 49 or more significant bits below the tile's bound, instead of once per addition.
// Why integers and not ds_add_f32: tools/probes/lds_atomic_probe.hip (profiles/r05_lds_atomic_probe.json) -- a wave64 ds_add_f32
// takes 192 CU cycles whatever its addresses (three per lane: the float LDS atomic is serialised), ds_add_u64 takes 6, ds_add_u32 4.
// Not covered (the caller falls back to plane_scatter_kernel): staging buffers of 2^32 floats or more, ray tiles that straddle items.

#ifndef GNERF_BIN_TILE
#define GNERF_BIN_TILE 16
#endif
constexpr int kBinTile = GNERF_BIN_TILE;                // texels per side of a plane tile (a power of two)
constexpr int kBinHalo = kBinTile + 1;                  // LDS tile side incl. the right / bottom halo
constexpr int kBinCountThreads = 256;
constexpr int kBinAccThreads = 1024;
constexpr int kBinAccWaves = kBinAccThreads / 64;
constexpr int kBinHaloTexels = 2 * kBinTile + 1;        // halo texels of a tile: right column (16), bottom row (16), corner
#ifndef GNERF_BIN_WALK_UNROLL
#define GNERF_BIN_WALK_UNROLL 1
#endif
constexpr int kBinWalkUnroll = GNERF_BIN_WALK_UNROLL;                       // samples per thread and trip of the walk kernels (loads of a trip issued together)
constexpr int kBinBatch = 4;                            // records per scalar-load batch; segments are padded to whole batches

struct BinRowCell { unsigned row, cell; };              // row: float index of the record's dX row in the staging buffer; cell: (ly * 17 + lx) * 32
typedef float BinWeights __attribute__((ext_vector_type(4)));      // the four tap weights x 2^min(s, 64): x0y0, x1y0, x0y1, x1y1

struct BinArgs {
    const float* stage;         // the staged dX rows (gnerf_render_backward's scatter_stage)
    float* grad_planes;
    int* count;                 // [n_ptiles] records per plane tile
    int* start;                 // [n_ptiles + 1] first record of the tile's segment (multiples of kBinBatch)
    int* cursor;                // [n_ptiles] fill position inside the segment
    unsigned* bound;            // [n_ptiles] float bits of max |dX| over the tile's records (non-negative floats order like unsigned)
    int* scale;                 // [n_ptiles] fixed-point exponent s of the tile (kBinFloat: accumulate in fp32 -- inf / NaN among the rows)
    int* order;                 // [n_ptiles] tiles in processing order (largest first);  order[n_ptiles] = the pull counter
    BinRowCell* rc;             // [capacity]
    BinWeights* wt;             // [capacity]
    float* halo;                // [n_ptiles][kBinHaloTexels][32]
    int n_ptiles, tiles_x, tiles_y, tiles_per_plane;
};
constexpr int kBinFloat = 0x7fffffff;

__host__ inline int64_t bin_capacity(int64_t rays, int n_all, int64_t n_ptiles) { return 3 * rays * n_all + (kBinBatch - 1) * n_ptiles + kBinBatch; }
// where the workspace starts: behind the staged rows and the spare line of gnerf_render_backward (256-byte aligned)
__host__ inline size_t bin_workspace_offset(int64_t rays, int n_all) {
    return (size_t(rays) * size_t(n_all) * 33 * sizeof(float) + 256 + 255) / 256 * 256;
}
// bytes of the workspace behind the staged rows (gnerf_render_backward_stage_bytes adds them)
__host__ inline size_t bin_workspace_bytes(int64_t rays, int n_all, int n_items, int H, int W) {
    const int64_t tx = (W + kBinTile - 1) / kBinTile, ty = (H + kBinTile - 1) / kBinTile, nt = int64_t(n_items) * 3 * tx * ty;
    auto up = [](int64_t b) { return (b + 255) / 256 * 256; };
    const int64_t cap = bin_capacity(rays, n_all, nt);
    return size_t(6 * up((nt + 1) * 4) + up(cap * int64_t(sizeof(BinRowCell))) + up(cap * int64_t(sizeof(BinWeights))) + up(nt * kBinHaloTexels * 32 * 4));
}
__host__ inline BinArgs bin_carve(char* base, const float* stage, float* grad, int64_t rays, int n_all, int n_items, int H, int W) {
    BinArgs A;
    A.tiles_x = (W + kBinTile - 1) / kBinTile; A.tiles_y = (H + kBinTile - 1) / kBinTile;
    A.tiles_per_plane = A.tiles_x * A.tiles_y;
    A.n_ptiles = n_items * 3 * A.tiles_per_plane;
    auto up = [](int64_t b) { return (b + 255) / 256 * 256; };
    const int64_t words = up((int64_t(A.n_ptiles) + 1) * 4), cap = bin_capacity(rays, n_all, A.n_ptiles);
    A.stage = stage; A.grad_planes = grad;
    A.count = reinterpret_cast<int*>(base);            base += words;
    A.start = reinterpret_cast<int*>(base);            base += words;
    A.cursor = reinterpret_cast<int*>(base);           base += words;
    A.bound = reinterpret_cast<unsigned*>(base);       base += words;
    A.scale = reinterpret_cast<int*>(base);            base += words;
    A.order = reinterpret_cast<int*>(base);            base += words;
    A.rc = reinterpret_cast<BinRowCell*>(base);        base += up(cap * int64_t(sizeof(BinRowCell)));
    A.wt = reinterpret_cast<BinWeights*>(base);        base += up(cap * int64_t(sizeof(BinWeights)));
    A.halo = reinterpret_cast<float*>(base);
    return A;
}

// The bilinear footprint of one (sample, plane): plane_taps' arithmetic (render_coop.inl), kept in the form the records store.
struct TapGeom { int x0, y0; float fx, fy; unsigned valid; };            // valid: bit 0 x0, 1 x1, 2 y0, 3 y1 inside the plane
__device__ __forceinline__ TapGeom tap_geom(int H, int W, float u, float v) {
    float ix = ((u + 1.f) * float(W) - 1.f) * 0.5f;
    float iy = ((v + 1.f) * float(H) - 1.f) * 0.5f;
    ix = clamp_nn(ix, -1.5f, float(W) + 0.5f);
    iy = clamp_nn(iy, -1.5f, float(H) + 0.5f);
    const float x0f = floorf(ix), y0f = floorf(iy);
    TapGeom g;
    g.fx = ix - x0f; g.fy = iy - y0f;
    g.x0 = int(x0f); g.y0 = int(y0f);
    const int x1 = g.x0 + 1, y1 = g.y0 + 1;
    g.valid = (g.x0 >= 0 && g.x0 < W ? 1u : 0u) | (x1 >= 0 && x1 < W ? 2u : 0u) | (g.y0 >= 0 && g.y0 < H ? 4u : 0u) | (y1 >= 0 && y1 < H ? 8u : 0u);
    return g;
}
// the four tap weights from (fx, fy, valid): the expressions of plane_taps, so that a record's contributions are the products the
// other scatter forms add (x0y0, x1y0, x0y1, x1y1)
__device__ __forceinline__ v4f tap_weights(float fx, float fy, unsigned valid) {
    const float wx0 = (valid & 1u) ? 1.f - fx : 0.f, wx1 = (valid & 2u) ? fx : 0.f;
    const float wy0 = (valid & 4u) ? (1.f - fy) * (1.f / 3.f) : 0.f, wy1 = (valid & 8u) ? fy * (1.f / 3.f) : 0.f;
    return (v4f){wx0 * wy0, wx1 * wy0, wx0 * wy1, wx1 * wy1};
}

__device__ __forceinline__ int64_t bin_tile_ray(const Params& P, int tile, int i) {        // ray i (0..15) of ray tile `tile`, or -1
    const gnerf_render_params& p = P.p;
    if (P.tiles_per_item > 0) {
        const int item = tile / P.tiles_per_item, tt = tile % P.tiles_per_item;
        const int tx = tt / P.tiles_y, ty = tt % P.tiles_y;
        return int64_t(item) * p.rays_per_item + int64_t(ty * 4 + (i >> 2)) * p.image_width + tx * 4 + (i & 3);
    }
    if (const int pad = linear_pad(P); pad > 0) {             // (ragged calls: pipe_seq_to_ray's padded order)
        const int64_t seq = int64_t(tile) * 16 + i;
        const int item = int(seq / pad), local = int(seq - int64_t(item) * pad);
        return (item < p.n_items && local < p.rays_per_item) ? int64_t(item) * p.rays_per_item + local : -1;
    }
    const int64_t ray = int64_t(tile) * 16 + i;
    return ray < P.total_rays ? ray : -1;
}

// One (sample, plane) of the walk shared by bin_count and bin_fill: plane tile (or -1: every tap has weight zero), the footprint's
// cell in the tile and its weights.  A footprint that starts one texel left of / above the plane (x0 = -1 or y0 = -1: those taps have
// weight zero) is moved onto the plane -- its x1 / y1 taps become the x0 / y0 taps -- so that all four cells of EVERY record lie
// inside the 17 x 17 LDS tile and the accumulation needs no validity test (a zero weight adds zero).
struct BinHit { int ptile; unsigned cell; v4f w; };
__device__ __forceinline__ BinHit bin_hit(const Params& P, const BinArgs& A, int item, int pl, float px, float py, float pz) {
    const float u = pl == 2 ? pz : px;
    const float v = pl == 0 ? py : (pl == 1 ? pz : px);
    const TapGeom g = tap_geom(P.p.plane_h, P.p.plane_w, u, v);
    BinHit h;
    h.w = tap_weights(g.fx, g.fy, g.valid);
    h.cell = 0;
    if (!(h.w[0] != 0.f || h.w[1] != 0.f || h.w[2] != 0.f || h.w[3] != 0.f)) { h.ptile = -1; return h; }       // (NaN weights count as taps: they reach the sum)
    const int cx = min(max(g.x0, 0), P.p.plane_w - 1), cy = min(max(g.y0, 0), P.p.plane_h - 1);
    const int tx = cx / kBinTile, ty = cy / kBinTile;
    h.ptile = (item * 3 + pl) * A.tiles_per_plane + ty * A.tiles_x + tx;
    int lx = g.x0 - tx * kBinTile, ly = g.y0 - ty * kBinTile;              // -1 .. 15
    if (lx < 0) { lx = 0; h.w[0] = h.w[1]; h.w[1] = 0.f; h.w[2] = h.w[3]; h.w[3] = 0.f; }
    if (ly < 0) { ly = 0; h.w[0] = h.w[2]; h.w[1] = h.w[3]; h.w[2] = 0.f; h.w[3] = 0.f; }
    h.cell = unsigned((ly * kBinHalo + lx) * 32);
    return h;
}

// the fixed-point exponent of a tile from its record count and the float bits of its bound
__device__ __forceinline__ int bin_scale_of(int n, unsigned bbits) {
    if (bbits >= 0x7f800000u) return kBinFloat;                               // inf / NaN among the rows
    if (bbits == 0 || n == 0) return 0;
    int e;
    (void)frexpf(__uint_as_float(bbits), &e);                                 // bound < 2^e
    const int nbits = n > 1 ? 32 - __clz(n - 1) : 0;                          // n <= 2^nbits
    return 61 - nbits - e;
}
// the part of the scale the stored weights carry (the rest, for bounds below 2^-3, multiplies dx in the accumulation)
__device__ __forceinline__ int bin_weight_scale(int s) { return s == kBinFloat ? 0 : min(s, 64); }

// ---- pass 1 and pass 3: FILL = false counts records and bounds per plane tile, FILL = true writes the records.
// LDS: [3 * tiles_per_plane] ints (this ray tile's records per plane tile of its item), then the same number of words for the bound
// (count pass) or the segment base reserved for this workgroup (fill pass), then 16 rays x 8 floats.
template <bool FILL>
__global__ __launch_bounds__(kBinCountThreads) void bin_walk_kernel(Params P, BinArgs A) {
    extern __shared__ __align__(16) float smem[];
    const gnerf_render_params& p = P.p;
    const int tid = threadIdx.x, n_loc = 3 * A.tiles_per_plane;
    int* hist = reinterpret_cast<int*>(smem);
    unsigned* aux = reinterpret_cast<unsigned*>(smem) + n_loc;
    float* rays = smem + 2 * n_loc;
    const int n_all = p.depth_resolution + p.depth_resolution_importance;
    const int tile = blockIdx.x;
    if (tid < 16) {
        const int64_t ray = bin_tile_ray(P, tile, tid);
        float* r = rays + tid * 8;
        if (ray >= 0) {
            r[0] = p.ray_origins[ray * 3 + 0]; r[1] = p.ray_origins[ray * 3 + 1]; r[2] = p.ray_origins[ray * 3 + 2];
            r[3] = p.ray_dirs[ray * 3 + 0];    r[4] = p.ray_dirs[ray * 3 + 1];    r[5] = p.ray_dirs[ray * 3 + 2];
        }
        reinterpret_cast<int*>(r)[6] = int(ray);
    }
    for (int i = tid; i < n_loc; i += kBinCountThreads) { hist[i] = 0; aux[i] = 0; }
    __syncthreads();
    const int first_ray = reinterpret_cast<const int*>(rays)[6];
    if (first_ray < 0) return;
    const int item = first_ray / p.rays_per_item;              // a ray tile never straddles items (checked by the launcher)
    const int loc0 = item * n_loc;                             // this item's plane tiles are [loc0, loc0 + n_loc)
    const int n_smp = 16 * n_all;

    // ---- sweep 1: this workgroup's records per plane tile (and, count pass, the bound: max |dX| of the rows behind them).
    // kBinWalkUnroll samples per thread at a time, every load of the group issued before the first is used.  (Measured with 1 / 2 / 3 / 6
    // in flight: no difference -- the count pass reads every dX row once, 0.8 GB in 0.18 ms = 4.6 TB/s, and the fill pass writes 0.55 GB of
    // records in 0.15 ms: both sit at the memory system's rate, not at its latency; profiles/r05_bin_walk_unroll.txt.  1 is shipped.)
    for (int s0 = tid; s0 < n_smp; s0 += kBinWalkUnroll * kBinCountThreads) {
        int rank[kBinWalkUnroll];
        const float* block[kBinWalkUnroll];
        const float* rayp[kBinWalkUnroll];
        float depth[kBinWalkUnroll];
#pragma unroll
        for (int u = 0; u < kBinWalkUnroll; u++) {
            const int s = s0 + u * kBinCountThreads;
            const int ri = min(s, n_smp - 1) / n_all;
            rank[u] = min(s, n_smp - 1) - ri * n_all;
            rayp[u] = rays + ri * 8;
            const int ray = s < n_smp ? reinterpret_cast<const int*>(rayp[u])[6] : -1;
            block[u] = ray >= 0 ? A.stage + int64_t(ray) * P.bwd_ray_stride : nullptr;
            depth[u] = block[u] ? block[u][rank[u]] : 0.f;
        }
        int hit[kBinWalkUnroll][3];
        bool any[kBinWalkUnroll];
#pragma unroll
        for (int u = 0; u < kBinWalkUnroll; u++) {
            const float* r = rayp[u];
            const float px = __fadd_rn(r[0], __fmul_rn(depth[u], r[3])) * P.box_scale;
            const float py = __fadd_rn(r[1], __fmul_rn(depth[u], r[4])) * P.box_scale;
            const float pz = __fadd_rn(r[2], __fmul_rn(depth[u], r[5])) * P.box_scale;
            any[u] = false;
#pragma unroll
            for (int pl = 0; pl < 3; pl++) {
                hit[u][pl] = block[u] ? bin_hit(P, A, item, pl, px, py, pz).ptile : -1;
                any[u] = any[u] || hit[u][pl] >= 0;
            }
        }
        unsigned bits[kBinWalkUnroll];
        if (!FILL) {
            v4f row[kBinWalkUnroll][8];
#pragma unroll
            for (int u = 0; u < kBinWalkUnroll; u++) {
                const v4f* src = reinterpret_cast<const v4f*>((any[u] ? block[u] : A.stage) + n_all + (any[u] ? rank[u] : 0) * 32);      // (no record: any valid row, unused)
#pragma unroll
                for (int q = 0; q < 8; q++) row[u][q] = src[q];
            }
#pragma unroll
            for (int u = 0; u < kBinWalkUnroll; u++) {
                float m = 0.f;
                bool nan = false;
#pragma unroll
                for (int q = 0; q < 8; q++)
#pragma unroll
                    for (int e = 0; e < 4; e++) { m = fmaxf(m, fabsf(row[u][q][e])); nan = nan || (row[u][q][e] != row[u][q][e]); }
                bits[u] = nan ? 0x7fc00000u : __float_as_uint(m);     // (a NaN orders above every finite bound and above +inf)
            }
        }
#pragma unroll
        for (int u = 0; u < kBinWalkUnroll; u++) {
            if (!any[u]) continue;
#pragma unroll
            for (int pl = 0; pl < 3; pl++)
                if (hit[u][pl] >= 0) {
                    atomicAdd(hist + (hit[u][pl] - loc0), 1);
                    if (!FILL) atomicMax(aux + (hit[u][pl] - loc0), bits[u]);
                }
        }
    }
    __syncthreads();
    // ---- publish: one global atomic per plane tile this ray tile touches
    for (int i = tid; i < n_loc; i += kBinCountThreads) {
        const int c = hist[i];
        if (c == 0) continue;
        if (!FILL) {
            atomicAdd(A.count + loc0 + i, c);
            atomicMax(A.bound + loc0 + i, aux[i]);
        } else {
            aux[i] = unsigned(A.start[loc0 + i] + atomicAdd(A.cursor + loc0 + i, c));      // this workgroup's run inside the tile's segment
            hist[i] = 0;                                                                    // becomes the position inside the run
        }
    }
    if (!FILL) return;
    __syncthreads();
    // ---- sweep 2 (fill pass): the records, the same kBinWalkUnroll samples per thread at a time
    for (int s0 = tid; s0 < n_smp; s0 += kBinWalkUnroll * kBinCountThreads) {
        int rank[kBinWalkUnroll];
        int64_t block[kBinWalkUnroll];
        const float* rayp[kBinWalkUnroll];
        float depth[kBinWalkUnroll];
#pragma unroll
        for (int u = 0; u < kBinWalkUnroll; u++) {
            const int s = s0 + u * kBinCountThreads;
            const int ri = min(s, n_smp - 1) / n_all;
            rank[u] = min(s, n_smp - 1) - ri * n_all;
            rayp[u] = rays + ri * 8;
            const int ray = s < n_smp ? reinterpret_cast<const int*>(rayp[u])[6] : -1;
            block[u] = ray >= 0 ? int64_t(ray) * P.bwd_ray_stride : -1;
            depth[u] = block[u] >= 0 ? A.stage[block[u] + rank[u]] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < kBinWalkUnroll; u++) {
            if (block[u] < 0) continue;
            const float* r = rayp[u];
            const float px = __fadd_rn(r[0], __fmul_rn(depth[u], r[3])) * P.box_scale;
            const float py = __fadd_rn(r[1], __fmul_rn(depth[u], r[4])) * P.box_scale;
            const float pz = __fadd_rn(r[2], __fmul_rn(depth[u], r[5])) * P.box_scale;
#pragma unroll
            for (int pl = 0; pl < 3; pl++) {
                const BinHit h = bin_hit(P, A, item, pl, px, py, pz);
                if (h.ptile < 0) continue;
                const int loc = h.ptile - loc0;
                const unsigned at = aux[loc] + unsigned(atomicAdd(hist + loc, 1));
                const int sw = bin_weight_scale(A.scale[h.ptile]);
                A.rc[at] = BinRowCell{unsigned(block[u] + n_all + int64_t(rank[u]) * 32), h.cell};
                A.wt[at] = (BinWeights){ldexpf(h.w[0], sw), ldexpf(h.w[1], sw), ldexpf(h.w[2], sw), ldexpf(h.w[3], sw)};
            }
        }
    }
}

// ---- pass 2 (one workgroup): segment starts = exclusive scan of the counts rounded up to whole batches, the padding records of every
// segment (row 0, weight zero: they add zero), every tile's scale, and the processing order: tiles by size class (floor(log2(count))
// + 1, largest first), so that the persistent workgroups of bin_accumulate do not end on the biggest tiles.  The order inside a class
// is whatever the atomics give: it schedules work, no result depends on it.
__global__ __launch_bounds__(1024) void bin_scan_kernel(BinArgs A) {
    __shared__ int wave_tot[16];
    __shared__ int class_n[33], class_at[33];
    __shared__ int carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid < 33) class_n[tid] = 0;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < A.n_ptiles; base += 1024) {
        const int i = base + tid;
        const int c = i < A.n_ptiles ? A.count[i] : 0;
        const int padded = (c + kBinBatch - 1) / kBinBatch * kBinBatch;
        int incl = padded;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int up = __shfl_up(incl, o); if (lane >= o) incl += up; }
        if (lane == 63) wave_tot[wv] = incl;
        __syncthreads();
        int before = carry_s;
        for (int w = 0; w < wv; w++) before += wave_tot[w];
        if (i < A.n_ptiles) {
            const int st = before + incl - padded;
            A.start[i] = st;
            A.scale[i] = bin_scale_of(c, A.bound[i]);
            for (int k = c; k < padded; k++) { A.rc[st + k] = BinRowCell{0u, 0u}; A.wt[st + k] = (BinWeights){0.f, 0.f, 0.f, 0.f}; }
            atomicAdd(class_n + (c > 0 ? 32 - __clz(c) : 0), 1);          // class 0: empty tiles, class k: 2^(k-1) <= count < 2^k
        }
        __syncthreads();
        if (tid == 1023) carry_s = before + incl;
        __syncthreads();
    }
    if (tid == 0) {
        A.start[A.n_ptiles] = carry_s;
        A.order[A.n_ptiles] = 0;                                          // the pull counter of bin_accumulate
        int at = 0;
        for (int k = 32; k >= 0; k--) { class_at[k] = at; at += class_n[k]; }
    }
    __syncthreads();
    for (int i = tid; i < A.n_ptiles; i += 1024) {
        const int c = A.count[i];
        A.order[atomicAdd(class_at + (c > 0 ? 32 - __clz(c) : 0), 1)] = i;
    }
}

// fp32 -> 64-bit fixed point: floor of t (|t| < 2^61) as {low word, high word}.
// Two things this leans on, both properties of the conversion instructions and both deterministic:
//  * lo = t - hi 2^32 lies in [0, 2^32] -- CLOSED at the top: for a small negative t (t = -1.5: hi = -1) the exact difference 2^32 - 1.5
//    is not a float and rounds up to 2^32.  v_cvt_u32_f32 saturates, so the low word becomes 0xFFFFFFFF and the sum is one unit of
//    2^-s (<= 2^-49 of the tile's bound) above floor(t); an explicit clamp to 4294967040.f would cost a half-rate instruction per
//    record and lane in the kernel's inner loop and give the same word.
//  * the conversions send NaN to 0.  The padding records of a segment (bin_scan_kernel) carry weight zero and point at row 0 of the
//    staging buffer -- the first ray's depths, not a row known to be finite; a non-finite depth there gives 0 * inf = NaN as the
//    contribution, which this conversion turns into the zero a padding record must add (such a ray's own records go down the fp32
//    ds_add_f32 path of its tile, where the NaN propagates as in the reference).
__device__ __forceinline__ unsigned long long bin_to_fixed(float t) {
    const float hi = floorf(t * 2.3283064365386963e-10f);                 // floor(t / 2^32): |hi| < 2^29, exact
    const float lo = fmaf(hi, -4294967296.f, t);                          // t - hi 2^32 (a 24-bit value minus its upper part; see above for the top end)
    const unsigned lo_u = __float2uint_rz(lo);
    const int hi_i = __float2int_rz(hi);
    return (static_cast<unsigned long long>(static_cast<unsigned>(hi_i)) << 32) | lo_u;
}
__device__ __forceinline__ float bin_from_fixed(unsigned long long v, int s) {
    return ldexpf(__ll2float_rn(static_cast<long long>(v)), -s);
}

// A batch of kBinBatch records through the scalar cache: four (row, cell) pairs and four weight quadruples are wave-uniform, so they
// live in SGPRs and cost no vector instruction.  hipcc knows nothing of loads inside an asm statement: the values may only be used
// behind bin_smem_wait, which names them as in/out operands.
typedef unsigned u8v __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
__device__ __forceinline__ u8v bin_sload_rc(const BinRowCell* p) { u8v v; asm volatile("s_load_dwordx8 %0, %1, 0x0" : "=s"(v) : "s"(p) : "memory"); return v; }
__device__ __forceinline__ f16v bin_sload_wt(const BinWeights* p) { f16v v; asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=s"(v) : "s"(p) : "memory"); return v; }
__device__ __forceinline__ void bin_smem_wait(u8v& a, f16v& b) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a), "+s"(b) :: "memory"); }

// ---- pass 4: persistent workgroups, one plane tile at a time.  (Its own small argument block: with the render kernels' Params in
// scalar registers next to the 56 of the record pipeline the compiler spilled a hundred of them to vector lanes.)
struct BinAccArgs {
    const float* stage; float* grad_planes; float* halo;
    const int* count; const int* start; const int* scale; const unsigned* bound; const int* order; int* pull;
    const BinRowCell* rc; const BinWeights* wt;
    int n_ptiles, tiles_x, tiles_per_plane, H, W;
    unsigned row_pitch, tex_pitch, plane_pitch;
};

// One record's two taps of this lane's footprint row, in fixed point.  d0 / d1: the lane's dX value for lanes of half 0 / half 1 and
// zero for the other half, so that a0 d0 + a2 d1 is EXACTLY the product with this half's weight (one term is an exact zero) -- the
// weights sit in scalar registers, and a per-half select between two of them costs two moves and a select per weight.
__device__ __forceinline__ void bin_add_fixed(unsigned long long* acc, int at, float a0, float a1, float a2, float a3, float d0, float d1) {
    __hip_atomic_fetch_add(acc + at, bin_to_fixed(__fmaf_rn(a0, d0, __fmul_rn(a2, d1))), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_add(acc + at + 32, bin_to_fixed(__fmaf_rn(a1, d0, __fmul_rn(a3, d1))), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

__global__ __launch_bounds__(kBinAccThreads, 8) void bin_accumulate_kernel(BinAccArgs A) {
    extern __shared__ __align__(16) unsigned long long acc[];             // [17][17][32], then one word: the tile this round works on
    // (all LDS of the kernel is dynamic: a static object on top of it would push the raised 160 KB limit over the hardware's)
    int& s_next = *reinterpret_cast<int*>(acc + kBinHalo * kBinHalo * 32);
    const int tid = threadIdx.x, lane = tid & 63, ch = lane & 31, half = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int kCells = kBinHalo * kBinHalo * 32;
    const int H = A.H, W = A.W;
    const int lane_cell = half * kBinHalo * 32 + ch;                       // half h adds the two taps of footprint row h
    const float m0 = half ? 0.f : 1.f, m1 = half ? 1.f : 0.f;
    for (;;) {
        if (tid == 0) s_next = atomicAdd(A.pull, 1);
        __syncthreads();
        const int q = __builtin_amdgcn_readfirstlane(s_next);
        if (q >= A.n_ptiles) break;
        const int t = __builtin_amdgcn_readfirstlane(A.order[q]);
        const int n = __builtin_amdgcn_readfirstlane(A.count[t]);
        if (n == 0) { __syncthreads(); continue; }                        // empty tile: nothing to add, its halo is never read (bin_halo checks the count)
        for (int i = tid; i < kCells; i += kBinAccThreads) acc[i] = 0;
        const int s = __builtin_amdgcn_readfirstlane(A.scale[t]);
        const bool as_float = s == kBinFloat;                             // inf / NaN among the rows: plain fp32 LDS adds (order-dependent; the values are garbage anyway)
        const int s_dx = as_float ? 0 : s - bin_weight_scale(s);          // what the stored weights do not carry (> 0 for bounds below 2^-3 only)
        const bool any = __builtin_amdgcn_readfirstlane(A.bound[t]) != 0; // (all rows zero: the sums are zero)
        const int nb = (n + kBinBatch - 1) / kBinBatch;                   // batches; wave w takes w, w + 16, ...
        const int first = __builtin_amdgcn_readfirstlane(A.start[t]);
        const BinRowCell* rc0 = A.rc + first;
        const BinWeights* wt0 = A.wt + first;
        __syncthreads();
        if (any && as_float) {
            float* accf = reinterpret_cast<float*>(acc);
            for (int i = wv; i < n; i += kBinAccWaves) {
                const BinRowCell r = rc0[i];
                const BinWeights w = wt0[i];
                const float d = (A.stage + r.row)[ch];
                const int at = int(r.cell) + lane_cell;
                unsafeAtomicAdd(accf + 2 * at, (half ? w[2] : w[0]) * d);
                unsafeAtomicAdd(accf + 2 * (at + 32), (half ? w[3] : w[1]) * d);
            }
        } else if (any && wv < nb) {
            auto batch_of = [&](int k) { return min(wv + k * kBinAccWaves, nb - 1) * kBinBatch; };     // (past the end: the last batch again, loaded and not used)
            // Three stages in flight: (row, cell) of batch k+2 and the weights of batch k+1 on their way through the scalar cache, the
            // four dX rows of batch k+1 on their way through the vector cache, batch k being added.
            // (two copies of the loop, chosen per tile: with the rescaling a run-time no-op the compiler keeps a v_ldexp_f32 and a select per
            //  record -- 2 of 21 vector instructions in a kernel whose time IS its vector instructions, profiles/r05_backward_kernel_counters.json)
            auto run = [&](auto rescale) {
                u8v rcA = bin_sload_rc(rc0 + batch_of(0)), rcB = bin_sload_rc(rc0 + batch_of(1));
                f16v wA = bin_sload_wt(wt0 + batch_of(0));
                bin_smem_wait(rcA, wA);
                asm volatile("" : "+s"(rcB));                                  // (behind the wait above: asm volatile statements keep their order)
                float dA[kBinBatch], dB[kBinBatch];
    #pragma unroll
                for (int u = 0; u < kBinBatch; u++) dA[u] = (A.stage + rcA[2 * u])[ch];
                for (int k = 0; wv + k * kBinAccWaves < nb; k++) {
                    u8v rcC = bin_sload_rc(rc0 + batch_of(k + 2));
                    f16v wB = bin_sload_wt(wt0 + batch_of(k + 1));
    #pragma unroll
                    for (int u = 0; u < kBinBatch; u++) dB[u] = (A.stage + rcB[2 * u])[ch];
    #pragma unroll
                    for (int u = 0; u < kBinBatch; u++) {
                        const float d = decltype(rescale)::value ? ldexpf(dA[u], s_dx) : dA[u];      // (s_dx = 0 unless the tile's bound is below 2^-3)
                        // (the four weights as scalars of their own: written `half ? wA[4u+2] : wA[4u]` the select becomes a DYNAMIC index into
                        //  the 16-register vector, which the compiler lowers to a chain of sixteen compares and selects per weight)
                        float a0 = wA[4 * u], a1 = wA[4 * u + 1], a2 = wA[4 * u + 2], a3 = wA[4 * u + 3];
                        asm volatile("" : "+s"(a0), "+s"(a1), "+s"(a2), "+s"(a3));
                        bin_add_fixed(acc, int(rcA[2 * u + 1]) + lane_cell, a0, a1, a2, a3, d * m0, d * m1);
                    }
                    bin_smem_wait(rcC, wB);                                     // (also retires this batch's LDS adds: one counter)
                    rcA = rcB; rcB = rcC; wA = wB;
    #pragma unroll
                    for (int u = 0; u < kBinBatch; u++) dA[u] = dB[u];
                }
            };
            if (s_dx != 0) run(std::true_type{}); else run(std::false_type{});
        }
        __syncthreads();
        // ---- the tile's own texels: added to grad_planes (plain load + store: this workgroup is the only writer of them)
        const int tpp = A.tiles_per_plane, ip = t / tpp, tt = t - ip * tpp;
        const int item = ip / 3, pl = ip - item * 3, ty = tt / A.tiles_x, tx = tt - ty * A.tiles_x;
        float* grad_item = A.grad_planes + int64_t(item) * 3 * H * W * 32;
        const float* accr = reinterpret_cast<const float*>(acc);
        for (int e = tid; e < kBinTile * kBinTile * 32; e += kBinAccThreads) {
            const int c2 = e & 31, lx = (e >> 5) & (kBinTile - 1), ly = e / (32 * kBinTile);
            const int x = tx * kBinTile + lx, y = ty * kBinTile + ly;
            if (x < W && y < H) {
                const int at = (ly * kBinHalo + lx) * 32 + c2;
                const float v = as_float ? accr[2 * at] : bin_from_fixed(acc[at], s);
                float* dst = grad_item + ((unsigned(y) * A.row_pitch + unsigned(x) * A.tex_pitch + unsigned(pl) * A.plane_pitch) >> 2) + c2;
                *dst += v;
            }
        }
        // ---- the halo (right column, bottom row, corner): to the side buffer, added by bin_halo to the neighbours' texels
        for (int e = tid; e < kBinHaloTexels * 32; e += kBinAccThreads) {
            const int c2 = e & 31, h = e >> 5;
            const int ly = h < kBinTile ? h : kBinTile, lx = h < kBinTile ? kBinTile : (h < 2 * kBinTile ? h - kBinTile : kBinTile);
            const int at = (ly * kBinHalo + lx) * 32 + c2;
            A.halo[(int64_t(t) * kBinHaloTexels + h) * 32 + c2] = as_float ? accr[2 * at] : bin_from_fixed(acc[at], s);
        }
        // (the barrier at the top of the next round separates these reads from its zero fill)
    }
}

// ---- pass 5: each plane tile adds, in a fixed order, what its left, upper and upper-left neighbours accumulated for its first
// column / first row.  One workgroup per tile; thread = (texel of the first column or row, channel).
__global__ __launch_bounds__(1024) void bin_halo_kernel(Params P, BinArgs A) {
    const gnerf_render_params& p = P.p;
    const int t = blockIdx.x, tid = threadIdx.x, c2 = tid & 31, k = tid >> 5;         // k: 0..15 first column (row k), 16..30 first row (column k - 15)
    if (k >= 2 * kBinTile - 1) return;
    const int H = p.plane_h, W = p.plane_w;
    const int tpp = A.tiles_per_plane, ip = t / tpp, tt = t - ip * tpp;
    const int item = ip / 3, pl = ip - item * 3, ty = tt / A.tiles_x, tx = tt - ty * A.tiles_x;
    const int lx = k < kBinTile ? 0 : k - (kBinTile - 1), ly = k < kBinTile ? k : 0;
    const int x = tx * kBinTile + lx, y = ty * kBinTile + ly;
    if (x >= W || y >= H) return;
    auto from = [&](int nt, int h) -> float {                               // halo texel h of tile nt, if that tile accumulated anything
        if (A.count[nt] == 0) return 0.f;
        return A.halo[(int64_t(nt) * kBinHaloTexels + h) * 32 + c2];
    };
    float add = 0.f;
    bool any = false;
    if (lx == 0 && tx > 0) { add += from(t - 1, ly); any = true; }                                  // left neighbour's right column
    if (ly == 0 && ty > 0) { add += from(t - A.tiles_x, kBinTile + lx); any = true; }               // upper neighbour's bottom row
    if (lx == 0 && ly == 0 && tx > 0 && ty > 0) { add += from(t - A.tiles_x - 1, 2 * kBinTile); any = true; }   // corner
    if (!any || add == 0.f) return;
    float* dst = A.grad_planes + int64_t(item) * 3 * H * W * 32 + ((unsigned(y) * P.row_pitch + unsigned(x) * P.tex_pitch + unsigned(pl) * P.plane_pitch) >> 2) + c2;
    *dst += add;
}

// The launch sequence.  `ws`: bin_workspace_bytes() behind the staged rows, zeroed here where it has to be.
static int launch_binned_scatter(const Params& P, float* stage, char* ws, float* grad_planes, hipStream_t s) {
    const gnerf_render_params& p = P.p;
    const int n_all = p.depth_resolution + p.depth_resolution_importance;
    BinArgs A = bin_carve(ws, stage, grad_planes, P.total_rays, n_all, p.n_items, p.plane_h, p.plane_w);
    const size_t head = reinterpret_cast<char*>(A.rc) - ws;               // counts, starts, cursors, bounds, scales, order
    if (hipMemsetAsync(ws, 0, head, s) != hipSuccess) return gnerf::fail(GNERF_E_LAUNCH, "render_backward: cannot clear the bin counters");
    const size_t lds_walk = (2 * size_t(3) * A.tiles_per_plane + 16 * 8) * sizeof(float);
    static PerDeviceOnce once_c, once_f, once_a;
    if (int e = once_c.raise_lds(bin_walk_kernel<false>, "render_backward")) return e;
    if (int e = once_f.raise_lds(bin_walk_kernel<true>, "render_backward")) return e;
    if (int e = once_a.raise_lds(bin_accumulate_kernel, "render_backward")) return e;
    hipLaunchKernelGGL(bin_walk_kernel<false>, dim3(P.n_tiles), dim3(kBinCountThreads), lds_walk, s, P, A);
    hipLaunchKernelGGL(bin_scan_kernel, dim3(1), dim3(1024), 0, s, A);
    hipLaunchKernelGGL(bin_walk_kernel<true>, dim3(P.n_tiles), dim3(kBinCountThreads), lds_walk, s, P, A);
    const int acc_blocks = min(A.n_ptiles, 2 * gnerf::kNumCU);
    BinAccArgs C;
    C.stage = A.stage; C.grad_planes = A.grad_planes; C.halo = A.halo;
    C.count = A.count; C.start = A.start; C.scale = A.scale; C.bound = A.bound; C.order = A.order; C.pull = A.order + A.n_ptiles;
    C.rc = A.rc; C.wt = A.wt;
    C.n_ptiles = A.n_ptiles; C.tiles_x = A.tiles_x; C.tiles_per_plane = A.tiles_per_plane; C.H = p.plane_h; C.W = p.plane_w;
    C.row_pitch = P.row_pitch; C.tex_pitch = P.tex_pitch; C.plane_pitch = P.plane_pitch;
    hipLaunchKernelGGL(bin_accumulate_kernel, dim3(acc_blocks), dim3(kBinAccThreads), size_t(kBinHalo) * kBinHalo * 32 * 8 + 16, s, C);
    hipLaunchKernelGGL(bin_halo_kernel, dim3(A.n_ptiles), dim3(1024), 0, s, P, A);
    return gnerf::check_launch("binned plane scatter");
}
