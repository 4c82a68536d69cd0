// v_cndmask_b32 issue cost in its different forms (issue_cost.hip measured 9.4 ns for the VOP2 form reading vcc).
#include <hip/hip_runtime.h>
#include <cstdio>
#define KERNEL(NAME, PRE, ASM, ...)                                                              \
    __global__ void NAME(float *out, int iters, float a, float b) {                              \
        float x[8]; for (int i = 0; i < 8; ++i) x[i] = threadIdx.x + i;                          \
        unsigned long long m = (threadIdx.x & 1) ? 0xAAAAAAAAAAAAAAAAull : 0x5555555555555555ull; \
        m = __builtin_amdgcn_readfirstlane((unsigned)m) | ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(m >> 32)) << 32); \
        for (int it = 0; it < iters; ++it) {                                                     \
            PRE;                                                                                 \
            _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                      \
                _Pragma("unroll") for (int j = 0; j < 8; ++j) { asm volatile(ASM : __VA_ARGS__); } \
            }                                                                                    \
        }                                                                                        \
        float s = 0; for (int i = 0; i < 8; ++i) s += x[i];                                      \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                          \
    }
KERNEL(k_e32_vcc, , "v_cndmask_b32 %0, %0, %1, vcc", "+v"(x[j]) : "v"(a))
KERNEL(k_e64_sgpr, , "v_cndmask_b32_e64 %0, %0, %1, %2", "+v"(x[j]) : "v"(a), "s"(m))
KERNEL(k_e64_const, , "v_cndmask_b32_e64 %0, 0, %1, %2", "+v"(x[j]) : "v"(a), "s"(m))
KERNEL(k_cmp_then, , "v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc", "+v"(x[j]) : "v"(a), "v"(b))
KERNEL(k_cmp64_then, , "v_cmp_lt_f32_e64 s[20:21], %0, %1\n v_cndmask_b32_e64 %0, %0, %2, s[20:21]", "+v"(x[j]) : "v"(a), "v"(b) : "s20", "s21")
KERNEL(k_bfi, , "v_bfi_b32 %0, %1, %0, %2", "+v"(x[j]) : "v"(a), "v"(b))
KERNEL(k_max, , "v_max_f32 %0, %0, %1", "+v"(x[j]) : "v"(a))
KERNEL(k_med3, , "v_med3_f32 %0, %0, %1, %2", "+v"(x[j]) : "v"(a), "v"(b))

template <typename K> void run(const char *name, K kern, float *d, int per) {
    const int iters = 4000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int thr : {512, 1024}) {
        hipLaunchKernelGGL(kern, dim3(256), dim3(thr), 0, 0, d, 10, 1.0001f, 0.5f);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(256), dim3(thr), 0, 0, d, iters, 1.0001f, 0.5f);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%-34s %d waves/SIMD: %.2f ns per asm statement per SIMD (%d instr each)\n", name, thr / 256, ms * 1e6 / ((double)iters * 32 * (thr / 256)), per);
    }
}
int main() {
    float *d; (void)hipMalloc(&d, 1 << 22);
    run("v_cndmask_b32 (e32, vcc)", k_e32_vcc, d, 1); run("v_cndmask_b32_e64 (sgpr pair)", k_e64_sgpr, d, 1);
    run("v_cndmask_b32_e64 0, v, sgpr", k_e64_const, d, 1); run("v_cmp vcc + v_cndmask vcc", k_cmp_then, d, 2);
    run("v_cmp_e64 s[] + v_cndmask_e64 s[]", k_cmp64_then, d, 2); run("v_bfi_b32", k_bfi, d, 1); run("v_max_f32", k_max, d, 1);
    run("v_med3_f32", k_med3, d, 1);
    return 0;
}
