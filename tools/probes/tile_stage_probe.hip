// north_star's "LDS-staged plane tiles", measured: does staging a 4x4-ray tile's texel footprint of one depth slab in LDS beat reading
// every bilinear tap straight from L1/L2?  Two kernels do the SAME work on config 2's geometry (4 items x 128x128 rays, 48 coarse
// depths, 3 x 32-channel 256x256 fp32 planes in the interleaved [N,H,W,96] layout, the bench's camera): per 16-sample slab (one sample
// of each ray of the tile at depth index k) tap records for 16 x 3 (sample, plane) pairs, then the 12-tap x 32-channel blend, 8 lanes
// per texel, exactly as the shipped kernel's lookup stage (render_coop.inl) -- reduced to one checksum per workgroup.
//   direct   24 global_load_dwordx4 per slab per lane group, addresses from the records                       (what ships)
//   staged   the slab's footprint per plane (bounding box of the 16 samples' 2x2 taps; <= 6x6 texels, else that slab falls back to
//            direct) is copied global -> LDS by the wave (8 lanes per 128-byte texel), and the 24 tap reads come from LDS
// Both checksums must agree to rounding (the compiler contracts the two blends differently).  Run under rocprofv3: kernel time from --kernel-trace --stats, L1->L2 requests from
// --pmc TCP_TCC_READ_REQ_sum (tools/probes/tile_stage_probe.sh -> gpurun_out/tile_stage_probe.json, kept as profiles/r03_tile_stage_probe.json).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));

struct Geo {
    const char* planes;      // [N][H][W][96] fp32
    float cam[4][12];        // per item: origin (3), then the three columns right / up / forward of cam2world
    float focal, box_scale, t0, dt;
    int H, W, res, S, n_items;
    unsigned tex_pitch, row_pitch, plane_pitch;
};

__device__ __forceinline__ void taps_of(int H, int W, float u, float v, unsigned tex_pitch, unsigned row_pitch, unsigned plane_off,
                                        uint4& off, v4f& wgt, int& x0o, int& y0o) {
    float ix = ((u + 1.f) * float(W) - 1.f) * 0.5f, iy = ((v + 1.f) * float(H) - 1.f) * 0.5f;
    ix = fminf(fmaxf(ix, -1.5f), float(W) + 0.5f);
    iy = fminf(fmaxf(iy, -1.5f), float(H) + 0.5f);
    const float x0f = floorf(ix), y0f = floorf(iy), fx = ix - x0f, fy = iy - y0f;
    const int x0 = int(x0f), y0 = int(y0f), x1 = x0 + 1, y1 = y0 + 1;
    const float wx0 = (x0 >= 0 && x0 < W) ? 1.f - fx : 0.f, wx1 = (x1 >= 0 && x1 < W) ? fx : 0.f;
    const float wy0 = (y0 >= 0 && y0 < H) ? (1.f - fy) * (1.f / 3.f) : 0.f, wy1 = (y1 >= 0 && y1 < H) ? fy * (1.f / 3.f) : 0.f;
    const int cx0 = min(max(x0, 0), W - 1), cx1 = min(max(x1, 0), W - 1), cy0 = min(max(y0, 0), H - 1), cy1 = min(max(y1, 0), H - 1);
    off = make_uint4(cy0 * row_pitch + cx0 * tex_pitch + plane_off, cy0 * row_pitch + cx1 * tex_pitch + plane_off,
                     cy1 * row_pitch + cx0 * tex_pitch + plane_off, cy1 * row_pitch + cx1 * tex_pitch + plane_off);
    wgt = (v4f){wx0 * wy0, wx1 * wy0, wx0 * wy1, wx1 * wy1};
    x0o = cx0; y0o = cy0;          // clamped corner (the footprint is taken over clamped coordinates: every tap address lies inside it)
}

constexpr int kBox = 6;                       // staged footprint: at most kBox x kBox texels per plane
constexpr int kRecDw = 24 + 8;                // per sample: 3 planes x (4 offsets + 4 weights), + per plane (cx0, cy0) and padding
constexpr int kWaveLds = 16 * kRecDw + 3 * kBox * kBox * 32 + 16;       // records, three staged footprints (32 floats per texel), box origins

template <bool STAGED>
__global__ __launch_bounds__(256, 2) void lookup_kernel(Geo G, float* out, unsigned* fallbacks, int wave_floats) {
    extern __shared__ __align__(16) float smem[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float* rec = smem + wv * wave_floats;
    float* box = rec + 16 * kRecDw;                       // [3][kBox*kBox][32]
    int* org = reinterpret_cast<int*>(box + 3 * kBox * kBox * 32);      // [3][4]: x0, y0, width, height of each plane's box
    // tile -> item, 4x4 pixel block (tiles walk down image columns like the render kernels)
    const int tiles_y = G.res / 4, tiles_per_item = tiles_y * tiles_y;
    const int tile = blockIdx.x, item = tile / tiles_per_item, tt = tile % tiles_per_item, tx = tt / tiles_y, ty = tt % tiles_y;
    const char* planes = G.planes + size_t(item) * G.H * G.W * 384;
    const float* cam = G.cam[item];
    v4f total = {0.f, 0.f, 0.f, 0.f};
    unsigned fell = 0;
    for (int k = wv; k < G.S; k += 4) {
        const float depth = G.t0 + G.dt * (float(k) + 0.37f);
        // ---- tap records: lane (sample j = ray of the tile, plane pl)
        int cx0 = 1 << 20, cy0 = 1 << 20;
        if (lane < 48) {
            const int j = lane & 15, pl = lane >> 4;
            const int px_i = tx * 4 + (j & 3), py_i = ty * 4 + (j >> 2);
            const float xc = ((float(px_i) + 0.5f) / float(G.res) - 0.5f) / G.focal, yc = ((float(py_i) + 0.5f) / float(G.res) - 0.5f) / G.focal;
            float dx = xc * cam[3] + yc * cam[6] + cam[9], dy = xc * cam[4] + yc * cam[7] + cam[10], dz = xc * cam[5] + yc * cam[8] + cam[11];
            const float inv = rsqrtf(dx * dx + dy * dy + dz * dz);
            dx *= inv; dy *= inv; dz *= inv;
            const float X = (cam[0] + depth * dx) * G.box_scale, Y = (cam[1] + depth * dy) * G.box_scale, Z = (cam[2] + depth * dz) * G.box_scale;
            const float u = pl == 2 ? Z : X, v = pl == 0 ? Y : (pl == 1 ? Z : X);
            uint4 off; v4f wgt;
            taps_of(G.H, G.W, u, v, G.tex_pitch, G.row_pitch, unsigned(pl) * G.plane_pitch, off, wgt, cx0, cy0);
            float* r = rec + j * kRecDw + pl * 8;
            *reinterpret_cast<uint4*>(r) = off;
            *reinterpret_cast<v4f*>(r + 4) = wgt;
            reinterpret_cast<int*>(rec + j * kRecDw + 24)[2 * pl] = cx0;
            reinterpret_cast<int*>(rec + j * kRecDw + 24)[2 * pl + 1] = cy0;
        }
        bool use_box = false;
        if (STAGED) {
            // ---- footprint per plane: min / max of the clamped corners over the 16 samples (DPP-free: plain shuffles, 4 steps in a row of 16)
            int xmin = lane < 48 ? cx0 : (1 << 20), ymin = lane < 48 ? cy0 : (1 << 20), xmax = lane < 48 ? cx0 : -1, ymax = lane < 48 ? cy0 : -1;
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) {
                xmin = min(xmin, __shfl_xor(xmin, o)); ymin = min(ymin, __shfl_xor(ymin, o));
                xmax = max(xmax, __shfl_xor(xmax, o)); ymax = max(ymax, __shfl_xor(ymax, o));
            }
            const int bw = xmax - xmin + 2, bh = ymax - ymin + 2;          // + the second tap column / row
            const bool fits = lane >= 48 || (bw <= kBox && bh <= kBox);
            use_box = __all(fits);
            if (lane < 48 && (lane & 15) == 0) { org[4 * (lane >> 4)] = xmin; org[4 * (lane >> 4) + 1] = ymin; org[4 * (lane >> 4) + 2] = bw; org[4 * (lane >> 4) + 3] = bh; }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (use_box) {
                // ---- copy the three footprints global -> LDS: 8 lanes per texel (16 bytes each), 8 texels per wave instruction
#pragma unroll
                for (int pl = 0; pl < 3; pl++) {
                    const int x0b = org[4 * pl], y0b = org[4 * pl + 1], bw2 = org[4 * pl + 2], bh2 = org[4 * pl + 3];
                    const int n_tex = bw2 * bh2;
                    for (int t = lane >> 3; t < n_tex; t += 8) {
                        const int yy = min(y0b + t / bw2, G.H - 1), xx = min(x0b + t % bw2, G.W - 1);
                        const v4f val = *reinterpret_cast<const v4f*>(planes + (unsigned(yy) * G.row_pitch + unsigned(xx) * G.tex_pitch + pl * G.plane_pitch) + (lane & 7) * 16);
                        *reinterpret_cast<v4f*>(box + (pl * kBox * kBox + t) * 32 + (lane & 7) * 4) = val;
                    }
                }
            } else if (lane == 0) {
                fell++;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // ---- the blend: 8 lanes per texel, 8 samples per step, two steps
        const int b = lane >> 3, cq16 = (lane & 7) * 16;
#pragma unroll
        for (int a = 0; a < 2; a++) {
            const float* r = rec + (8 * a + b) * kRecDw;
            v4f acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int pl = 0; pl < 3; pl++) {
                const uint4 off = *reinterpret_cast<const uint4*>(r + pl * 8);
                const v4f wgt = *reinterpret_cast<const v4f*>(r + pl * 8 + 4);
                v4f t00, t01, t10, t11;
                if (STAGED && use_box) {
                    const int cx = reinterpret_cast<const int*>(r + 24)[2 * pl], cy = reinterpret_cast<const int*>(r + 24)[2 * pl + 1];
                    const int bw2 = org[4 * pl + 2];
                    const int lx = cx - org[4 * pl], ly = cy - org[4 * pl + 1];
                    // the second tap column / row: +1 unless the corner was clamped at the plane's far edge (then the same texel, as in `off`)
                    const int dx1 = off.y != off.x ? 1 : 0, dy1 = off.z != off.x ? bw2 : 0;
                    const float* base = box + (pl * kBox * kBox + ly * bw2 + lx) * 32 + (cq16 >> 2);
                    t00 = *reinterpret_cast<const v4f*>(base);
                    t01 = *reinterpret_cast<const v4f*>(base + dx1 * 32);
                    t10 = *reinterpret_cast<const v4f*>(base + dy1 * 32);
                    t11 = *reinterpret_cast<const v4f*>(base + (dy1 + dx1) * 32);
                } else {
                    t00 = *reinterpret_cast<const v4f*>(planes + off.x + cq16);
                    t01 = *reinterpret_cast<const v4f*>(planes + off.y + cq16);
                    t10 = *reinterpret_cast<const v4f*>(planes + off.z + cq16);
                    t11 = *reinterpret_cast<const v4f*>(planes + off.w + cq16);
                }
                acc += t00 * wgt[0] + t01 * wgt[1] + t10 * wgt[2] + t11 * wgt[3];
            }
            total += acc;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // the records and the boxes are rewritten by the next slab
    }
    // checksum: per wave, lane-wise sums into out[tile][wave][lane][4]
    *reinterpret_cast<v4f*>(out + ((size_t(tile) * 4 + wv) * 64 + lane) * 4) = total;
    if (STAGED && lane == 0 && fell) atomicAdd(fallbacks, fell);
}

int main(int argc, char** argv) {
    const char* which = argc > 1 ? argv[1] : "both";
    const int N = 4, H = 256, W = 256, res = 128, S = 48;
    const size_t plane_bytes = size_t(N) * H * W * 96 * 4;
    std::vector<float> hp(plane_bytes / 4);
    unsigned st = 12345u;
    for (auto& v : hp) { st = st * 1664525u + 1013904223u; v = float(int(st >> 8) - (1 << 23)) * (1.f / float(1 << 22)); }
    char* d_planes; float *d_out_a, *d_out_b; unsigned* d_fb;
    const int tiles = N * (res / 4) * (res / 4);
    const size_t out_floats = size_t(tiles) * 4 * 64 * 4;
    hipMalloc(&d_planes, plane_bytes); hipMalloc(&d_out_a, out_floats * 4); hipMalloc(&d_out_b, out_floats * 4); hipMalloc(&d_fb, 4);
    hipMemcpy(d_planes, hp.data(), plane_bytes, hipMemcpyHostToDevice);
    hipMemset(d_fb, 0, 4);
    Geo G;
    G.planes = d_planes; G.H = H; G.W = W; G.res = res; G.S = S; G.n_items = N;
    G.tex_pitch = 384; G.row_pitch = 384 * W; G.plane_pitch = 128;
    G.focal = 4.2647f; G.box_scale = 2.f; G.t0 = 2.25f; G.dt = (3.3f - 2.25f) / 47.f;
    for (int i = 0; i < N; i++) {               // LookAtPoseSampler.sample(3.14/2 + 0.1 i, 3.14/2, radius 2.7): camera on the +z side looking at the origin
        const float theta = 3.14f / 2 + 0.1f * i, phi = 3.14f / 2, r = 2.7f;
        const float o[3] = {r * sinf(phi) * cosf(3.14159265f - theta), r * cosf(phi), r * sinf(phi) * sinf(3.14159265f - theta)};
        float f[3] = {-o[0], -o[1], -o[2]};
        const float fn = sqrtf(f[0] * f[0] + f[1] * f[1] + f[2] * f[2]);
        for (float& v : f) v /= fn;
        float rt[3] = {-(1.f * f[2] - 0.f * f[1]), -(0.f * f[0] - 0.f * f[2]), -(0.f * f[1] - 1.f * f[0])};       // -cross(up, f), up = (0,1,0)
        const float rn = sqrtf(rt[0] * rt[0] + rt[1] * rt[1] + rt[2] * rt[2]);
        for (float& v : rt) v /= rn;
        const float up[3] = {f[1] * rt[2] - f[2] * rt[1], f[2] * rt[0] - f[0] * rt[2], f[0] * rt[1] - f[1] * rt[0]};
        const float vals[12] = {o[0], o[1], o[2], rt[0], rt[1], rt[2], up[0], up[1], up[2], f[0], f[1], f[2]};
        memcpy(G.cam[i], vals, sizeof(vals));
    }
    const size_t lds = 4 * kWaveLds * sizeof(float);
    hipFuncSetAttribute(reinterpret_cast<const void*>(lookup_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
    hipFuncSetAttribute(reinterpret_cast<const void*>(lookup_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms_a = 0, ms_b = 0, ms_c = 0;
    const int reps = 20;
    const int small_wave = 16 * kRecDw + 16;
    const size_t lds_small = 4 * small_wave * sizeof(float);                  // the direct kernel needs only the tap records: 8 KB per workgroup
    if (strcmp(which, "staged")) {
        for (int i = 0; i < 3; i++) hipLaunchKernelGGL(lookup_kernel<false>, dim3(tiles), dim3(256), lds, 0, G, d_out_a, d_fb, kWaveLds);
        hipEventRecord(e0);
        for (int i = 0; i < reps; i++) hipLaunchKernelGGL(lookup_kernel<false>, dim3(tiles), dim3(256), lds, 0, G, d_out_a, d_fb, kWaveLds);
        hipEventRecord(e1); hipDeviceSynchronize(); hipEventElapsedTime(&ms_a, e0, e1); ms_a /= reps;
        // third arm: the direct kernel at the occupancy it allows by itself (8 KB of LDS per workgroup instead of the staged kernel's 64 KB)
        for (int i = 0; i < 3; i++) hipLaunchKernelGGL(lookup_kernel<false>, dim3(tiles), dim3(256), lds_small, 0, G, d_out_a, d_fb, small_wave);
        hipEventRecord(e0);
        for (int i = 0; i < reps; i++) hipLaunchKernelGGL(lookup_kernel<false>, dim3(tiles), dim3(256), lds_small, 0, G, d_out_a, d_fb, small_wave);
        hipEventRecord(e1); hipDeviceSynchronize(); hipEventElapsedTime(&ms_c, e0, e1); ms_c /= reps;
    }
    if (strcmp(which, "direct")) {
        for (int i = 0; i < 3; i++) hipLaunchKernelGGL(lookup_kernel<true>, dim3(tiles), dim3(256), lds, 0, G, d_out_b, d_fb, kWaveLds);
        hipMemset(d_fb, 0, 4);
        hipEventRecord(e0);
        for (int i = 0; i < reps; i++) hipLaunchKernelGGL(lookup_kernel<true>, dim3(tiles), dim3(256), lds, 0, G, d_out_b, d_fb, kWaveLds);
        hipEventRecord(e1); hipDeviceSynchronize(); hipEventElapsedTime(&ms_b, e0, e1); ms_b /= reps;
    }
    if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "kernel failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
    unsigned fb = 0; hipMemcpy(&fb, d_fb, 4, hipMemcpyDeviceToHost);
    long mism = -1;
    if (!strcmp(which, "both")) {
        std::vector<float> a(out_floats), b(out_floats);
        hipMemcpy(a.data(), d_out_a, out_floats * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), d_out_b, out_floats * 4, hipMemcpyDeviceToHost);
        // (the two instantiations contract the blend's multiply-adds differently, so the sums agree to rounding, not bit for bit)
        mism = 0;
        for (size_t i = 0; i < out_floats; i++) { const double d = fabs(double(a[i]) - double(b[i])); if (d > 1e-4 * (1.0 + fabs(double(a[i])))) mism++; }
    }
    const double slabs = double(tiles) * S;
    printf("{\"workload\": \"config-2 coarse samples: %d tiles of 4x4 rays x %d depth slabs, 12 taps x 32 channels per sample\", \"direct_ms_at_the_staged_kernels_occupancy\": %.4f, \"direct_ms_at_its_own_occupancy\": %.4f, \"staged_ms\": %.4f, "
           "\"staged_slabs_that_fell_back_to_direct\": %.4f, \"checksum_entries_off_by_more_than_1e-4_relative\": %ld, \"tap_bytes\": %.0f, \"lds_bytes_per_workgroup\": %zu}\n",
           tiles, S, ms_a, ms_c, ms_b, fb / double(reps) / slabs, mism, slabs * 16 * 12 * 128, lds);
    return mism > 0 ? 2 : 0;
}
