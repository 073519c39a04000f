// Where do the workgroups (and their four waves) of a 768 x 256-thread launch with 46 KB of LDS land?
// Prints XCC / SE / CU per workgroup and the SIMD of each of its waves.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256, 3) void census(unsigned* out, int spin) {
    extern __shared__ float smem[];
    smem[threadIdx.x] = threadIdx.x;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);      // HW_REG_HW_ID, all 32 bits
        unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);    // HW_REG_XCC_ID[3:0]
        out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = hw; out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = xcc;
    }
    // keep the workgroup resident for a while so that all 768 coexist
    float acc = smem[threadIdx.x];
    for (int i = 0; i < spin; i++) acc = acc * 1.0001f + 0.5f;
    if (acc == 123.456f) out[0] = 0;
}
int main() {
    const int G = 768;
    unsigned* d; hipMalloc(&d, G * 8 * sizeof(unsigned));
    hipFuncSetAttribute((const void*)census, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    hipLaunchKernelGGL(census, dim3(G), dim3(256), 46 * 1024, 0, d, 200000);
    std::vector<unsigned> h(G * 8);
    hipMemcpy(h.data(), d, G * 8 * sizeof(unsigned), hipMemcpyDeviceToHost);
    for (int b = 0; b < G; b++) {
        unsigned hw = h[b * 8], xcc = h[b * 8 + 1];
        printf("%d xcc=%u se=%u sh=%u cu=%u simd=", b, xcc & 15, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15);
        for (int w = 0; w < 4; w++) printf("%u", (h[(b * 4 + w) * 2] >> 4) & 3);
        printf(" waveslot=");
        for (int w = 0; w < 4; w++) printf("%u,", h[(b * 4 + w) * 2] & 15);
        printf("\n");
    }
    return 0;
}
