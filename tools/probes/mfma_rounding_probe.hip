// How does v_mfma_f32_16x16x4_f32 round its accumulation?  c = 1, a = 1, b = {1.5 * 2^-24, 0, 0, 0}: round-to-nearest gives 1 + 2^-23, truncation 1.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ void k(float* out, float b0, float b1, float b2, float b3, float c0) {
    const int lane = threadIdx.x, g = lane >> 4;
    const float b[4] = {b0, b1, b2, b3};
    v4f c = {c0, c0, c0, c0};
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, b[g], c, 0, 0, 0);      // A[i][k] = 1, B[k][j] = b[k] (k = lane >> 4)
    out[lane] = c[0];
    float s = c0;                       // the same sum with IEEE round-to-nearest FMAs, in k order
    for (int q = 0; q < 4; q++) s = fmaf(1.0f, b[q], s);
    if (lane == 0) out[64] = s;
}
int main() {
    float* d; hipMalloc(&d, 65 * 4); float h[65];
    const float e = ldexpf(1.f, -24);
    struct { float b[4], c; const char* what; } t[] = {
        {{1.5f * e, 0, 0, 0}, 1.f, "1 + 1.5 ulp/2 (RNE: 1+2^-23, RTZ: 1)"},
        {{0.75f * e, 0.75f * e, 0, 0}, 1.f, "1 + 0.75 + 0.75 half-ulps (exact sum 1 + 1.5 half-ulps; sequential RNE: 1+2^-23 after 2nd? )"},
        {{0.6f * e, 0.6f * e, 0.6f * e, 0.6f * e}, 1.f, "four terms of 0.6 half-ulp (exact 1 + 2.4 half-ulps -> 1+2^-23; sequential RNE: each rounds to ... )"},
        {{-1.5f * e, 0, 0, 0}, 1.f, "1 - 1.5 half-ulp"},
        {{3.f, 5.f, 7.f, 11.f}, 0.f, "plain 26"}};
    for (auto& c : t) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, c.b[0], c.b[1], c.b[2], c.b[3], c.c);
        hipMemcpy(h, d, 65 * 4, hipMemcpyDeviceToHost);
        printf("%-90s mfma %.9g (%a)   fma chain %.9g (%a)\n", c.what, h[0], h[0], h[64], h[64]);
    }
    return 0;
}
