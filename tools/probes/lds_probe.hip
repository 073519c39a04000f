// Does SQ_LDS_BANK_CONFLICT count the extra passes of a conflict-free wide LDS read?  Three kernels do the same number of
// ds_read_b128 per lane with different address patterns; run each under `rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS`:
//   linear   lane i reads 16 bytes at 16 i            (conflict-free by any definition: the 64 lanes cover 1 KB contiguously)
//   pitch36  lane (j = i & 15, g = i >> 4) reads 16 bytes at (36 j + 8 g) dwords   (the render kernel's staging-row read)
//   same_bank every lane reads 16 bytes at 256 i bytes (all lanes on banks 0..3: a true 64-way conflict)
//   b32_linear lane i reads 4 bytes at 4 i, 4 times     (narrow conflict-free reads, for the baseline ratio)
//   gen P Q A B   lane i reads 16 bytes at (P * (i & A) + Q * (i >> B)) dwords: scan candidate pitches of a layout
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/lds_probe.hip -o /tmp/lds_probe ; run: /tmp/lds_probe <linear|pitch36|same_bank|b32_linear>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdlib>
typedef float v4f __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(64) void probe(float* out, int iters, int P = 0, int Q = 0, int A = 15, int B = 4) {
    __shared__ __align__(16) float lds[64 * 64 + 64];
    (void)P; (void)Q;
    const int lane = threadIdx.x;
    for (int i = lane; i < 64 * 64 + 64; i += 64) lds[i] = float(i);
    __syncthreads();
    int off;                                                   // in dwords
    if (MODE == 0) off = 4 * lane;
    else if (MODE == 1) off = 36 * (lane & 15) + 8 * (lane >> 4);
    else if (MODE == 2) off = 64 * lane;
    else if (MODE == 4) off = (P * (lane & A) + Q * (lane >> B)) & 4092;
    else off = lane;
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; it++) {
        if (MODE == 3) {
#pragma unroll
            for (int k = 0; k < 4; k++) { float v; asm volatile("ds_read_b32 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(off * 4), "n"(0)); acc[k] += v; }
        } else {
            v4f v;
            asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(off * 4));
            acc += v;
        }
    }
    out[blockIdx.x * 64 + lane] = acc[0] + acc[1] + acc[2] + acc[3];
}
int main(int argc, char** argv) {
    const char* m = argc > 1 ? argv[1] : "linear";
    float* out; hipMalloc(&out, 1024 * 64 * sizeof(float));
    const int iters = 4096;
    for (int rep = 0; rep < 3; rep++) {
        if (!strcmp(m, "linear")) hipLaunchKernelGGL(probe<0>, dim3(1024), dim3(64), 0, 0, out, iters);
        else if (!strcmp(m, "pitch36")) hipLaunchKernelGGL(probe<1>, dim3(1024), dim3(64), 0, 0, out, iters);
        else if (!strcmp(m, "same_bank")) hipLaunchKernelGGL(probe<2>, dim3(1024), dim3(64), 0, 0, out, iters);
        else if (!strcmp(m, "gen")) hipLaunchKernelGGL(probe<4>, dim3(1024), dim3(64), 0, 0, out, iters, atoi(argv[2]), atoi(argv[3]), argc > 4 ? atoi(argv[4]) : 15, argc > 5 ? atoi(argv[5]) : 4);
        else hipLaunchKernelGGL(probe<3>, dim3(1024), dim3(64), 0, 0, out, iters);
    }
    hipDeviceSynchronize();
    printf("%s done\n", m);
    return 0;
}
