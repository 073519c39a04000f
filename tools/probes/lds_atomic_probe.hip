// LDS atomic rates on gfx950: how many CU cycles does a wave64 ds_add_{u32,f32,u64} cost, by address pattern and waves per CU?
// (Sizing the backward's plane-gradient accumulation in LDS: DESIGN.md section 3.2, round 5.)
//   patterns   linear   lane i adds at element i                       (one 256-byte / 512-byte run)
//              rows2    lanes 0-31 at row A + lane, lanes 32-63 at row B + lane (two 128-byte texel rows: the accumulate pattern)
//              same     every lane at element 0                        (64-way same-address)
// One workgroup of 64 W threads per CU (grid = 256); every wave runs ITER x 8 atomics between two s_memtime reads.
// Reported: CU cycles per wave-instruction = slowest wave's cycles x 1 / (ITER x 8 x W) ... i.e. aggregate over the W waves of the CU.
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/lds_atomic_probe.hip -o /tmp/lds_atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

constexpr int ITER = 512;
template <int OP, int PAT>
__global__ __launch_bounds__(1024) void probe(unsigned long long* out, int rowB) {
    extern __shared__ __align__(16) unsigned long long lds[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int i = tid; i < 8192; i += blockDim.x) lds[i] = 0;
    __syncthreads();
    const int esz = OP == 2 ? 8 : 4;
    int elem;
    if (PAT == 0) elem = lane + 64 * wv;
    else if (PAT == 1) elem = (lane < 32 ? (37 * wv) % 97 * 32 : rowB * 32 + (53 * wv) % 89 * 32) + (lane & 31);
    else elem = 0;
    const unsigned addr = unsigned(elem * esz);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; it++) {
        if (OP == 0) {
            asm volatile("ds_add_u32 %0, %1\n\tds_add_u32 %0, %1\n\tds_add_u32 %0, %1\n\tds_add_u32 %0, %1\n\t"
                         "ds_add_u32 %0, %1\n\tds_add_u32 %0, %1\n\tds_add_u32 %0, %1\n\tds_add_u32 %0, %1" :: "v"(addr), "v"(1u) : "memory");
        } else if (OP == 1) {
            asm volatile("ds_add_f32 %0, %1\n\tds_add_f32 %0, %1\n\tds_add_f32 %0, %1\n\tds_add_f32 %0, %1\n\t"
                         "ds_add_f32 %0, %1\n\tds_add_f32 %0, %1\n\tds_add_f32 %0, %1\n\tds_add_f32 %0, %1" :: "v"(addr), "v"(1.0f) : "memory");
        } else {
            const unsigned long long one = 1;
            asm volatile("ds_add_u64 %0, %1\n\tds_add_u64 %0, %1\n\tds_add_u64 %0, %1\n\tds_add_u64 %0, %1\n\t"
                         "ds_add_u64 %0, %1\n\tds_add_u64 %0, %1\n\tds_add_u64 %0, %1\n\tds_add_u64 %0, %1" :: "v"(addr), "v"(one) : "memory");
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[blockIdx.x * 16 + wv] = t1 - t0;
    __syncthreads();
    if (tid == 0 && lds[0] == 0xdeadbeefull) out[0] = 0;       // keep the LDS live
}

template <int OP, int PAT>
double run(int W, unsigned long long* d) {
    hipFuncSetAttribute((const void*)probe<OP, PAT>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL((probe<OP, PAT>), dim3(256), dim3(64 * W), 65536, 0, d, 128);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256 * 16);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> per_cu;
    for (int b = 0; b < 256; b++) {
        unsigned long long mx = 0;
        for (int w = 0; w < W; w++) mx = std::max(mx, h[b * 16 + w]);
        per_cu.push_back(double(mx) / (double(ITER) * 8 * W));
    }
    std::sort(per_cu.begin(), per_cu.end());
    return per_cu[128];
}

int main() {
    unsigned long long* d; hipMalloc(&d, 256 * 16 * 8);
    const char* ops[3] = {"ds_add_u32", "ds_add_f32", "ds_add_u64"};
    const char* pats[3] = {"linear", "rows2", "same"};
    printf("{\"unit\": \"CU cycles per wave64 instruction (median CU; slowest wave of the CU / instructions of all its waves)\", \"rows\": [\n");
    bool first = true;
    for (int W : {1, 4, 8, 16}) {
        double r[3][3] = {{run<0, 0>(W, d), run<0, 1>(W, d), run<0, 2>(W, d)}, {run<1, 0>(W, d), run<1, 1>(W, d), run<1, 2>(W, d)}, {run<2, 0>(W, d), run<2, 1>(W, d), run<2, 2>(W, d)}};
        for (int o = 0; o < 3; o++)
            for (int p = 0; p < 3; p++) {
                printf("%s {\"op\": \"%s\", \"pattern\": \"%s\", \"waves_per_cu\": %d, \"cu_cycles_per_instruction\": %.2f}", first ? "" : ",\n", ops[o], pats[p], W, r[o][p]);
                first = false;
            }
    }
    printf("\n]}\n");
    return 0;
}
