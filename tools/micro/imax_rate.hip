// Issue cost of integer min / max / compare on gfx950 against the float ones (cycles a wave64 instruction occupies
// its SIMD): plane distances that matter to a slab test are non-negative floats, which order like integers.
// Build: hipcc -O3 --offload-arch=gfx950 imax_rate.hip -o imax_rate
#include <hip/hip_runtime.h>
#include <cstdio>

#define K(NAME, ASM)                                                                                  \
    __global__ __launch_bounds__(256) void NAME(unsigned* out, int iters, unsigned a, unsigned b) {   \
        unsigned x0 = threadIdx.x | 0x3c003c00u, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7; \
        for (int i = 0; i < iters; i++) {                                                             \
            _Pragma("unroll") for (int r = 0; r < 8; r++) {                                           \
                asm volatile(ASM : "+v"(x0) : "v"(a), "v"(b) : "vcc"); asm volatile(ASM : "+v"(x1) : "v"(a), "v"(b) : "vcc"); \
                asm volatile(ASM : "+v"(x2) : "v"(a), "v"(b) : "vcc"); asm volatile(ASM : "+v"(x3) : "v"(a), "v"(b) : "vcc"); \
                asm volatile(ASM : "+v"(x4) : "v"(a), "v"(b) : "vcc"); asm volatile(ASM : "+v"(x5) : "v"(a), "v"(b) : "vcc"); \
                asm volatile(ASM : "+v"(x6) : "v"(a), "v"(b) : "vcc"); asm volatile(ASM : "+v"(x7) : "v"(a), "v"(b) : "vcc"); \
            }                                                                                         \
        }                                                                                             \
        out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;                  \
    }
K(k_max_f32, "v_max_f32 %0, %0, %1")
K(k_max_i32, "v_max_i32 %0, %0, %1")
K(k_max_u32, "v_max_u32 %0, %0, %1")
K(k_min_i32, "v_min_i32 %0, %0, %1")
K(k_max3_f32, "v_max3_f32 %0, %0, %1, %2")
K(k_max3_i32, "v_max3_i32 %0, %0, %1, %2")
K(k_min3_u32, "v_min3_u32 %0, %0, %1, %2")
K(k_med3_i32, "v_med3_i32 %0, %0, %1, %2")
K(k_cmp_f32, "v_cmp_le_f32 vcc, %0, %1")
K(k_cmp_i32, "v_cmp_le_i32 vcc, %0, %1")
K(k_cmp_u32, "v_cmp_le_u32 vcc, %0, %1")
K(k_add_u32, "v_add_u32 %0, %0, %1")
K(k_sub_f32, "v_sub_f32 %0, %0, %1")

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
    printf("%-14s %8.3f ms  %5.2f cycles per wave instruction at 2.4 GHz\n", name, best, best * 1e-3 * 2.4e9 / (8.0 * iters * 64.0));
}

int main() {
    unsigned* out; (void)hipMalloc(&out, 256 * 8 * 256 * sizeof(unsigned));
    run("v_add_u32", k_add_u32, out); run("v_sub_f32", k_sub_f32, out);
    run("v_max_f32", k_max_f32, out); run("v_max_i32", k_max_i32, out); run("v_max_u32", k_max_u32, out); run("v_min_i32", k_min_i32, out);
    run("v_max3_f32", k_max3_f32, out); run("v_max3_i32", k_max3_i32, out); run("v_min3_u32", k_min3_u32, out); run("v_med3_i32", k_med3_i32, out);
    run("v_cmp_le_f32", k_cmp_f32, out); run("v_cmp_le_i32", k_cmp_i32, out); run("v_cmp_le_u32", k_cmp_u32, out);
    (void)hipFree(out);
    return 0;
}
