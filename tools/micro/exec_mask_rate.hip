// Does a wave64 VALU instruction cost less when part of the wave is masked off?  Times a long chain-free
// v_fma_f32 loop under different EXEC masks: all 64 lanes, the low 32, the low 16, the high 32, every other lane,
// one lane.  (If the SIMD skipped a 32-lane pass whose lanes are all off, packing a wave's live lanes into one half
// would pay; if every mask costs the same, only the lane count per instruction matters and nothing is to be
// gained from where the live lanes sit.)
// Build: hipcc -O3 --offload-arch=gfx950 exec_mask_rate.hip -o exec_mask_rate
#include <hip/hip_runtime.h>
#include <cstdio>

#define OP(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b))
#define OPD(x) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(da), "v"(db))
#define OPM(x) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b))

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b, unsigned long long mask) {
    const int lane = threadIdx.x & 63;
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    double d0 = x0, d1 = x1, d2 = x2, d3 = x3; const double da = a, db = b;
    if ((mask >> lane) & 1ull) {
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int r = 0; r < 8; r++) {
                if (KIND == 0) { OP(x0); OP(x1); OP(x2); OP(x3); OP(x4); OP(x5); OP(x6); OP(x7); }
                if (KIND == 1) { OPD(d0); OPD(d1); OPD(d2); OPD(d3); OPD(d0); OPD(d1); OPD(d2); OPD(d3); }
                if (KIND == 2) { OPM(x0); OPM(x1); OPM(x2); OPM(x3); OPM(x4); OPM(x5); OPM(x6); OPM(x7); }
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + (float)(d0 + d1 + d2 + d3);
}

template <int KIND>
static void run(const char* name, float* out, unsigned long long mask, const char* what) {
    const int iters = 20000, blocks = 256 * 8;            // 8 workgroups of 4 waves per CU: 8 waves per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f, mask);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    // wave instructions per SIMD: 8 waves x iters x 64
    const double insts = 8.0 * iters * 64.0;
    printf("%-10s %-26s %8.3f ms  %5.2f cycles per wave instruction at 2.4 GHz\n", name, what, best, best * 1e-3 * 2.4e9 / insts);
}

int main() {
    float* out; hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    struct { unsigned long long m; const char* what; } masks[] = {
        {~0ull, "all 64 lanes"}, {0xffffffffull, "lanes 0-31"}, {0xffffffff00000000ull, "lanes 32-63"}, {0xffffull, "lanes 0-15"},
        {0x5555555555555555ull, "every other lane"}, {0x00000000ffff0000ull, "lanes 16-31"}, {1ull, "lane 0"}, {0x0000ffff0000ffffull, "lanes 0-15 and 32-47"}};
    for (auto& m : masks) run<0>("v_fma_f32", out, m.m, m.what);
    for (auto& m : masks) run<1>("v_fma_f64", out, m.m, m.what);
    for (auto& m : masks) run<2>("v_max3_f32", out, m.m, m.what);
    hipFree(out);
    return 0;
}
