// Does v_mfma_f32_16x16x32_f16 keep f16 subnormal inputs on gfx950?  (decides the hi/lo split form)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(float a_val, float b_val, float *out) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)a_val; b[i] = (_Float16)b_val; }
    f4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    out[threadIdx.x] = c[0];
}
int main() {
    float *d; hipMalloc(&d, 256);
    const float cases[][2] = {{1.f, 9.5367431640625e-07f /*2^-20*/}, {9.5367431640625e-07f, 1.f},
                              {1.f, 5.9604644775390625e-08f /*2^-24 smallest*/}, {0.00048828125f, 0.00048828125f /*2^-11 x 2^-11*/},
                              {3.0517578125e-05f /*2^-15*/, 3.0517578125e-05f}};
    for (auto &c : cases) {
        hipLaunchKernelGGL(k, 1, 64, 0, 0, c[0], c[1], d);
        float h[64]; hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
        printf("a=%g b=%g  mfma sum over K=32: %g   expected %g\n", c[0], c[1], h[0], 32.0 * (double)c[0] * (double)c[1]);
    }
    return 0;
}
