// Issue cost per instruction type as the pooling kernel sees it: two waves per SIMD (512 threads per CU), independent
// instructions (8 rotating register sets), ns per instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

#define KERNEL(NAME, ASM, ...)                                                                   \
    __global__ void NAME(float *out, int iters, float a, float b) {                              \
        float x[8]; f2 p[8]; unsigned u[8];                                                      \
        for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x + i; p[i] = f2{x[i], x[i] + 1}; u[i] = threadIdx.x * 7 + i; } \
        for (int it = 0; it < iters; ++it) {                                                     \
            _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                      \
                _Pragma("unroll") for (int j = 0; j < 8; ++j) { asm volatile(ASM : __VA_ARGS__); } \
            }                                                                                    \
        }                                                                                        \
        float s = 0; for (int i = 0; i < 8; ++i) s += x[i] + p[i].x + p[i].y + (float)u[i];      \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                          \
    }

KERNEL(k_fma, "v_fma_f32 %0, %0, %1, %2", "+v"(x[j]) : "v"(a), "v"(b))
KERNEL(k_mul, "v_mul_f32 %0, %0, %1", "+v"(x[j]) : "v"(a))
KERNEL(k_fmac, "v_fmac_f32 %0, %1, %2", "+v"(x[j]) : "v"(a), "v"(b))
KERNEL(k_pkfma, "v_pk_fma_f32 %0, %0, %1, %1", "+v"(p[j]) : "v"(p[(j + 1) & 7]))
KERNEL(k_pkmul, "v_pk_mul_f32 %0, %0, %1", "+v"(p[j]) : "v"(p[(j + 1) & 7]))
KERNEL(k_pkadd, "v_pk_add_f32 %0, %0, %1", "+v"(p[j]) : "v"(p[(j + 1) & 7]))
KERNEL(k_cvtpk, "v_cvt_pkrtz_f16_f32 %0, %0, %1", "+v"(x[j]) : "v"(a))
KERNEL(k_cvt16, "v_cvt_f16_f32 %0, %0", "+v"(x[j]) : "v"(a))
KERNEL(k_mix, "v_fma_mix_f32 %0, %0, %1, %2 op_sel_hi:[0,0,1]", "+v"(x[j]) : "v"(a), "v"(b))
KERNEL(k_mixlo, "v_fma_mixlo_f16 %0, %1, %2, %0", "+v"(x[j]) : "v"(a), "v"(b))
KERNEL(k_cnd, "v_cndmask_b32 %0, %0, %1, vcc", "+v"(x[j]) : "v"(a))
KERNEL(k_mov, "v_mov_b32 %0, %1", "+v"(x[j]) : "v"(a))
KERNEL(k_xor, "v_xor_b32 %0, %0, %1", "+v"(u[j]) : "v"(u[(j + 1) & 7]))
KERNEL(k_bitop3, "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96", "+v"(u[j]) : "v"(u[(j + 1) & 7]), "v"(u[(j + 2) & 7]))
KERNEL(k_sqrt, "v_sqrt_f32 %0, %0", "+v"(x[j]) : "v"(a))
KERNEL(k_rcp, "v_rcp_f32 %0, %0", "+v"(x[j]) : "v"(a))
KERNEL(k_cmp, "v_cmp_lt_f32 vcc, %0, %1", "+v"(x[j]) : "v"(a))
KERNEL(k_perm, "v_perm_b32 %0, %0, %1, %2", "+v"(u[j]) : "v"(u[(j + 1) & 7]), "v"(u[(j + 2) & 7]))
KERNEL(k_nop, "s_nop 0", "+v"(x[j]) : "v"(a))

template <typename K>
void run(const char *name, K kern, float *d) {
    const int iters = 4000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int thr : {256, 512, 1024}) {
        hipLaunchKernelGGL(kern, dim3(256), dim3(thr), 0, 0, d, 10, 1.0001f, 0.5f);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(256), dim3(thr), 0, 0, d, iters, 1.0001f, 0.5f);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double n = (double)iters * 32 * (thr / 256);
        printf("%-22s %d waves/SIMD: %.2f ns per instruction per SIMD\n", name, thr / 256, ms * 1e6 / n);
    }
}

int main() {
    float *d; (void)hipMalloc(&d, 1 << 22);
    run("v_fma_f32", k_fma, d); run("v_mul_f32", k_mul, d); run("v_fmac_f32", k_fmac, d);
    run("v_pk_fma_f32", k_pkfma, d); run("v_pk_mul_f32", k_pkmul, d); run("v_pk_add_f32", k_pkadd, d);
    run("v_cvt_pkrtz_f16_f32", k_cvtpk, d); run("v_cvt_f16_f32", k_cvt16, d); run("v_fma_mix_f32", k_mix, d);
    run("v_fma_mixlo_f16", k_mixlo, d); run("v_cndmask_b32", k_cnd, d); run("v_mov_b32", k_mov, d);
    run("v_xor_b32", k_xor, d); run("v_bitop3_b32", k_bitop3, d); run("v_sqrt_f32", k_sqrt, d); run("v_rcp_f32", k_rcp, d);
    run("v_cmp_lt_f32", k_cmp, d); run("v_perm_b32", k_perm, d); run("s_nop 0", k_nop, d);
    return 0;
}
