#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
template <int HI> __device__ __forceinline__ v2f pk_mul_bc(v2f t, v2f w) {
    v2f r;
    if (HI) asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=&v"(r) : "v"(t), "v"(w));
    else    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=&v"(r) : "v"(t), "v"(w));
    return r;
}
template <int HI> __device__ __forceinline__ v2f pk_fma_bc(v2f t, v2f w, v2f c) {
    if (HI) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(c) : "v"(t), "v"(w));
    else    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "+v"(c) : "v"(t), "v"(w));
    return c;
}
__global__ void k(const v4f* tex, const v4f* wgt, v4f* out_asm, v4f* out_c, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    v4f t[4]; for (int q = 0; q < 4; q++) t[q] = tex[4 * i + q];
    const v4f w = wgt[i];
    const v2f w01 = {w[0], w[1]}, w23 = {w[2], w[3]};
    v2f s[2];
    for (int q = 0; q < 2; q++) {
        const v2f t0 = {t[0][2*q], t[0][2*q+1]}, t1 = {t[1][2*q], t[1][2*q+1]}, t2 = {t[2][2*q], t[2][2*q+1]}, t3 = {t[3][2*q], t[3][2*q+1]};
        s[q] = pk_fma_bc<1>(t3, w23, pk_fma_bc<0>(t2, w23, pk_fma_bc<1>(t1, w01, pk_mul_bc<0>(t0, w01))));
    }
    out_asm[i] = (v4f){s[0][0], s[0][1], s[1][0], s[1][1]};
    v4f r;
    for (int c = 0; c < 4; c++) r[c] = __fmaf_rn(t[3][c], w[3], __fmaf_rn(t[2][c], w[2], __fmaf_rn(t[1][c], w[1], __fmul_rn(t[0][c], w[0]))));
    out_c[i] = r;
}
int main() {
    const int n = 1 << 20;
    float* h = (float*)malloc(n * 20 * 4);
    srand(1);
    for (int i = 0; i < n * 20; i++) { float v = (rand() / float(RAND_MAX)) * 4.f - 2.f; if (rand() % 17 == 0) v = 0.f; h[i] = v; }
    v4f *tex, *wgt, *oa, *oc;
    hipMalloc(&tex, n * 64); hipMalloc(&wgt, n * 16); hipMalloc(&oa, n * 16); hipMalloc(&oc, n * 16);
    hipMemcpy(tex, h, n * 64, hipMemcpyHostToDevice); hipMemcpy(wgt, h + n * 16, n * 16, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, tex, wgt, oa, oc, n);
    float* a = (float*)malloc(n * 16), *c = (float*)malloc(n * 16);
    hipMemcpy(a, oa, n * 16, hipMemcpyDeviceToHost); hipMemcpy(c, oc, n * 16, hipMemcpyDeviceToHost);
    long bad = 0; double worst = 0;
    for (long i = 0; i < (long)n * 4; i++) if (a[i] != c[i]) { bad++; double d = a[i] - c[i]; if (d < 0) d = -d; if (d > worst) worst = d; if (bad < 5) printf("i=%ld asm=%g c=%g\n", i, a[i], c[i]); }
    printf("mismatches %ld of %ld, worst %g\n", bad, (long)n * 4, worst);
    return 0;
}
