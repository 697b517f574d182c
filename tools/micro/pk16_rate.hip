// Issue cost of packed half-precision VALU instructions on gfx950 (cycles a wave64 instruction occupies its
// SIMD, 8 waves per SIMD, 8 independent chains): would a slab test of BOTH child boxes in packed fp16
// (conservatively widened) cost less than two tests in fp32?
// Build: hipcc -O3 --offload-arch=gfx950 pk16_rate.hip -o pk16_rate
#include <hip/hip_runtime.h>
#include <cstdio>

#define K(NAME, ASM)                                                                                  \
    __global__ __launch_bounds__(256) void NAME(unsigned* out, int iters, unsigned a, unsigned b) {   \
        unsigned x0 = threadIdx.x | 0x3c003c00u, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7; \
        for (int i = 0; i < iters; i++) {                                                             \
            _Pragma("unroll") for (int r = 0; r < 8; r++) {                                           \
                asm volatile(ASM : "+v"(x0) : "v"(a), "v"(b)); asm volatile(ASM : "+v"(x1) : "v"(a), "v"(b)); \
                asm volatile(ASM : "+v"(x2) : "v"(a), "v"(b)); asm volatile(ASM : "+v"(x3) : "v"(a), "v"(b)); \
                asm volatile(ASM : "+v"(x4) : "v"(a), "v"(b)); asm volatile(ASM : "+v"(x5) : "v"(a), "v"(b)); \
                asm volatile(ASM : "+v"(x6) : "v"(a), "v"(b)); asm volatile(ASM : "+v"(x7) : "v"(a), "v"(b)); \
            }                                                                                         \
        }                                                                                             \
        out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;                  \
    }
K(k_pk_fma, "v_pk_fma_f16 %0, %0, %1, %2")
K(k_pk_mul, "v_pk_mul_f16 %0, %0, %1")
K(k_pk_add, "v_pk_add_f16 %0, %0, %1")
K(k_pk_max, "v_pk_max_f16 %0, %0, %1")
K(k_pk_min, "v_pk_min_f16 %0, %0, %1")
K(k_fma32, "v_fma_f32 %0, %0, %1, %2")
K(k_max32, "v_max_f32 %0, %0, %1")
K(k_cvt16, "v_cvt_f16_f32 %0, %0")
K(k_cvtpk, "v_cvt_pkrtz_f16_f32 %0, %0, %1")
K(k_pkfma32, "v_fma_f32 %0, %0, %1, %2")
K(k_max3_16, "v_max3_f16 %0, %0, %1, %2")
K(k_perm, "v_perm_b32 %0, %0, %1, %2")
K(k_cmp16, "v_cmp_le_f16 vcc, %0, %1")

template <class F>
static void run(const char* name, F f, unsigned* out) {
    const int iters = 20000, blocks = 256 * 8;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(f, dim3(blocks), dim3(256), 0, 0, out, iters, 0x3c003c01u, 0x38003801u);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    printf("%-22s %8.3f ms  %5.2f cycles per wave instruction at 2.4 GHz\n", name, best, best * 1e-3 * 2.4e9 / (8.0 * iters * 64.0));
}

int main() {
    unsigned* out; (void)hipMalloc(&out, 256 * 8 * 256 * sizeof(unsigned));
    run("v_fma_f32", k_fma32, out); run("v_max_f32", k_max32, out);
    run("v_pk_fma_f16", k_pk_fma, out); run("v_pk_mul_f16", k_pk_mul, out); run("v_pk_add_f16", k_pk_add, out);
    run("v_pk_max_f16", k_pk_max, out); run("v_pk_min_f16", k_pk_min, out); run("v_max3_f16", k_max3_16, out);
    run("v_cvt_f16_f32", k_cvt16, out); run("v_cvt_pkrtz_f16_f32", k_cvtpk, out); run("v_perm_b32", k_perm, out);
    run("v_cmp_le_f16", k_cmp16, out);
    (void)hipFree(out);
    return 0;
}
