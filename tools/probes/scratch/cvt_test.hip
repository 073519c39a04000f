#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float* x, unsigned* out) {
    const int i = threadIdx.x;
    float a = x[2 * i], b = x[2 * i + 1];
    unsigned hi, lo_new, lo_old, t;
    float da, db;
    asm volatile("v_cvt_pk_f16_f32 %0, %5, %6\n\t"
                 "v_fma_mix_f32 %3, %0, -1.0, %5 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
                 "v_fma_mix_f32 %4, %0, -1.0, %6 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
                 "v_cvt_pk_f16_f32 %1, %3, %4\n\t"
                 "v_fma_mixlo_f16 %2, %0, -1.0, %5 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
                 "v_fma_mixhi_f16 %2, %0, -1.0, %6 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
                 : "=&v"(hi), "=&v"(lo_new), "=&v"(lo_old), "=&v"(da), "=&v"(db) : "v"(a), "v"(b));
    out[4 * i] = hi; out[4 * i + 1] = lo_new; out[4 * i + 2] = lo_old; out[4 * i + 3] = __float_as_uint(da);
}
int main() {
    float hx[8] = {0.1f, 1.0003f, 0.01f, 3.14159f, 1e-3f, 0.12f, 100.03f, 6e-5f};
    float* dx; unsigned* dout; unsigned ho[16];
    hipMalloc(&dx, sizeof(hx)); hipMalloc(&dout, sizeof(ho));
    hipMemcpy(dx, hx, sizeof(hx), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(4), 0, 0, dx, dout);
    hipMemcpy(ho, dout, sizeof(ho), hipMemcpyDeviceToHost);
    for (int i = 0; i < 4; i++) printf("x=(%g,%g) hi=%08x lo_new=%08x lo_old=%08x d=%g\n", hx[2*i], hx[2*i+1], ho[4*i], ho[4*i+1], ho[4*i+2], *(float*)&ho[4*i+3]);
    return 0;
}
