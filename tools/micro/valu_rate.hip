// Issue cost of individual gfx950 VALU instructions: cycles a wave64 instruction occupies its
// SIMD when 8 independent chains per wave and several waves per SIMD keep the pipe full.
// Build: hipcc -O3 --offload-arch=gfx950 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2v __attribute__((ext_vector_type(2)));

#define KERNEL(NAME, DECL, OP)                                                                       \
    __global__ __launch_bounds__(256) void NAME(float* out, int iters, float a, float b) {           \
        DECL;                                                                                        \
        for (int i = 0; i < iters; i++) {                                                            \
            _Pragma("unroll") for (int r = 0; r < 8; r++) { OP(x0); OP(x1); OP(x2); OP(x3); OP(x4); OP(x5); OP(x6); OP(x7); } \
        }                                                                                            \
        float acc = 0; SUM;                                                                          \
        out[blockIdx.x * 256 + threadIdx.x] = acc;                                                   \
    }

#define DECL_F float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7
#define DECL_P float2v x0 = {(float)threadIdx.x, 1}, x1 = {2, 3}, x2 = {4, 5}, x3 = {6, 7}, x4 = {8, 9}, x5 = {1, 3}, x6 = {5, 7}, x7 = {2, 9}; float2v pa = {a, a}, pb = {b, b}
#define DECL_D double x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7; double da = a, db = b
#define DECL_I unsigned x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7; unsigned ia = (unsigned)a + 3, ib = (unsigned)b + 5

#define SUM acc = (float)x0 + (float)x1 + (float)x2 + (float)x3 + (float)x4 + (float)x5 + (float)x6 + (float)x7
#define A3(op) asm volatile(op " %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b))
#define OP_FMA(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b))
#define OP_ADD(x) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(a))
#define OP_MUL(x) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(a))
#define OP_MIN(x) asm volatile("v_min_f32 %0, %0, %1" : "+v"(x) : "v"(a))
#define OP_MAX3(x) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b))
#define OP_MED3(x) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b))
#define OP_CND(x) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(a))
#define OP_CND3(x) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[22:23]" : "+v"(x) : "v"(a))
#define OP_CNDI(x) asm volatile("v_cndmask_b32 %0, 0, %0, vcc" : "+v"(x))
#define OP_MAX(x) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x) : "v"(a))
#define OP_SUB(x) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x) : "v"(a))
#define OP_FMAC(x) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x) : "v"(a), "v"(b))
#define OP_FMAS(x) asm volatile("v_fma_f32 %0, %0, s24, s24" : "+v"(x))
#define OP_MINS(x) asm volatile("v_min_f32 %0, s24, %0" : "+v"(x))
#define OP_CNDW(x) asm volatile("v_cmp_le_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(a) : "vcc")
#define OP_CNDW2(x) asm volatile("v_cmp_le_f32 vcc, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(a) : "vcc")
#define OP_CNDS(x) asm volatile("v_cmp_le_f32 s[26:27], %0, %1\n\tv_cndmask_b32_e64 %0, %0, %1, s[26:27]" : "+v"(x) : "v"(a) : "s26", "s27")
#define OP_VMOVS(x) asm volatile("v_mov_b32 %0, s24" : "+v"(x))
#define OP_CMP(x) asm volatile("v_cmp_le_f32 vcc, %0, %1" : : "v"(x), "v"(a) : "vcc")
#define OP_CMPS(x) asm volatile("v_cmp_le_f32 s[20:21], %0, %1" : : "v"(x), "v"(a) : "s20", "s21")
#define OP_MOV(x) asm volatile("v_mov_b32 %0, %1" : "+v"(x) : "v"(a))
#define OP_FMAMK(x) asm volatile("v_fmamk_f32 %0, %0, 0x3f800004, %1" : "+v"(x) : "v"(a))
#define OP_FMAINL(x) asm volatile("v_fma_f32 %0, %0, 1.0, %1" : "+v"(x) : "v"(a))
#define OP_ADDLIT(x) asm volatile("v_add_f32 %0, 0x3f800004, %0" : "+v"(x))
#define OP_MAX3S(x) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b))
#define OP_RCP(x) asm volatile("v_rcp_f32 %0, %0" : "+v"(x))
#define OP_SQRT(x) asm volatile("v_sqrt_f32 %0, %0" : "+v"(x))
KERNEL(k_fma, DECL_F, OP_FMA) KERNEL(k_add, DECL_F, OP_ADD) KERNEL(k_mul, DECL_F, OP_MUL) KERNEL(k_min, DECL_F, OP_MIN)
KERNEL(k_max3, DECL_F, OP_MAX3) KERNEL(k_med3, DECL_F, OP_MED3) KERNEL(k_cnd, DECL_F, OP_CND) KERNEL(k_cmp, DECL_F, OP_CMP)
KERNEL(k_cnd3, DECL_F, OP_CND3) KERNEL(k_cndi, DECL_F, OP_CNDI) KERNEL(k_max, DECL_F, OP_MAX) KERNEL(k_sub, DECL_F, OP_SUB) KERNEL(k_fmac, DECL_F, OP_FMAC)
KERNEL(k_fmas, DECL_F, OP_FMAS) KERNEL(k_mins, DECL_F, OP_MINS)
KERNEL(k_cndw, DECL_F, OP_CNDW) KERNEL(k_cndw2, DECL_F, OP_CNDW2) KERNEL(k_cnds, DECL_F, OP_CNDS) KERNEL(k_vmovs, DECL_F, OP_VMOVS)
KERNEL(k_cmps, DECL_F, OP_CMPS) KERNEL(k_mov, DECL_F, OP_MOV) KERNEL(k_rcp, DECL_F, OP_RCP) KERNEL(k_sqrt, DECL_F, OP_SQRT)
#undef SUM
#define SUM acc = x0.x + x1.y + x2.x + x3.y + x4.x + x5.y + x6.x + x7.y
#define OP_PKFMA(x) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(pa), "v"(pb))
#define OP_PKMUL(x) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x) : "v"(pa))
#define OP_PKADD(x) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x) : "v"(pa))
KERNEL(k_pkfma, DECL_P, OP_PKFMA) KERNEL(k_pkmul, DECL_P, OP_PKMUL) KERNEL(k_pkadd, DECL_P, OP_PKADD)
#undef SUM
#define SUM acc = (float)(x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7)
#define OP_DFMA(x) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(da), "v"(db))
#define OP_DMUL(x) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x) : "v"(da))
#define OP_DADD(x) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x) : "v"(da))
KERNEL(k_dfma, DECL_D, OP_DFMA) KERNEL(k_dmul, DECL_D, OP_DMUL) KERNEL(k_dadd, DECL_D, OP_DADD)
#define OP_IADD(x) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(ia))
#define OP_AND(x) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x) : "v"(ia))
#define OP_XOR3(x) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x) : "v"(ia))
#define OP_LSHLADD(x) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(x) : "v"(ia))
#define OP_MULLO(x) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(ia))
#define OP_MULHI(x) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x) : "v"(ia))
#define OP_ALIGN(x) asm volatile("v_alignbit_b32 %0, %0, %0, 13" : "+v"(x))
#define OP_MAD64(x) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(x) : "v"(ia), "v"(ib) : "vcc")
#define DECL_L unsigned long long x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7; unsigned ia = (unsigned)a + 3, ib = (unsigned)b + 5
KERNEL(k_mad64, DECL_L, OP_MAD64)
#define OP_CVT(x) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(x))
#define OP_CVTW(x) asm volatile("v_cvt_f32_u32_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "+v"(x))
KERNEL(k_cvt, DECL_I, OP_CVT) KERNEL(k_cvtw, DECL_I, OP_CVTW)
KERNEL(k_iadd, DECL_I, OP_IADD) KERNEL(k_and, DECL_I, OP_AND) KERNEL(k_xor, DECL_I, OP_XOR3) KERNEL(k_lshladd, DECL_I, OP_LSHLADD)
KERNEL(k_mullo, DECL_I, OP_MULLO) KERNEL(k_mulhi, DECL_I, OP_MULHI) KERNEL(k_align, DECL_I, OP_ALIGN)

KERNEL(k_fmamk, DECL_F, OP_FMAMK) KERNEL(k_fmainl, DECL_F, OP_FMAINL) KERNEL(k_addlit, DECL_F, OP_ADDLIT)

typedef void (*kern_t)(float*, int, float, float);
void run(const char* name, kern_t k, int waves_per_simd, double flops_per_inst) {
    int iters = 4000, blocks = 256 * waves_per_simd;
    float* out;
    (void)hipMalloc(&out, sizeof(float) * blocks * 256);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 10, 1.0001f, 0.5f);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    double insts = (double)iters * 64 * waves_per_simd;        // per SIMD
    printf("%-12s %d waves/SIMD  %7.3f ms  %5.2f cycles/inst @2.4GHz", name, waves_per_simd, ms, ms * 1e-3 * 2.4e9 / insts);
    if (flops_per_inst > 0) printf("  %6.1f TFLOP/s", flops_per_inst * 64 * insts * 1024 / (ms * 1e-3) / 1e12);
    printf("\n");
    (void)hipFree(out);
}

int main() {
    const int w = 8;
    run("v_fma_f32", k_fma, w, 2); run("v_fma_f32", k_fma, 2, 2); run("v_add_f32", k_add, w, 1); run("v_mul_f32", k_mul, w, 1);
    run("v_min_f32", k_min, w, 1); run("v_max3_f32", k_max3, w, 2); run("v_med3_f32", k_med3, w, 2);
    run("cmp+cnd vcc (2 inst)", k_cndw, w, 0); run("cmp+add+cnd vcc (3)", k_cndw2, w, 0); run("cmp+cnd sgpr (2 inst)", k_cnds, w, 0); run("v_mov v,s", k_vmovs, w, 0);
    run("v_cndmask", k_cnd, w, 0); run("v_cndmask e64 sgpr", k_cnd3, w, 0); run("v_cndmask 0,v", k_cndi, w, 0); run("v_max_f32", k_max, w, 1); run("v_sub_f32", k_sub, w, 1);
    run("v_fmac_f32", k_fmac, w, 2); run("v_fma sgpr ops", k_fmas, w, 2); run("v_min sgpr", k_mins, w, 1); run("v_cmp->vcc", k_cmp, w, 0); run("v_cmp->sgpr", k_cmps, w, 0); run("v_mov_b32", k_mov, w, 0);
    run("v_fmamk literal", k_fmamk, w, 2); run("v_fma inline 1.0", k_fmainl, w, 2); run("v_add literal", k_addlit, w, 1);
    run("v_rcp_f32", k_rcp, w, 1); run("v_sqrt_f32", k_sqrt, w, 1);
    run("v_pk_fma_f32", k_pkfma, w, 4); run("v_pk_mul_f32", k_pkmul, w, 2); run("v_pk_add_f32", k_pkadd, w, 2);
    run("v_fma_f64", k_dfma, w, 2); run("v_mul_f64", k_dmul, w, 1); run("v_add_f64", k_dadd, w, 1);
    run("v_add_u32", k_iadd, w, 0); run("v_and_b32", k_and, w, 0); run("v_xor_b32", k_xor, w, 0); run("v_lshl_add", k_lshladd, w, 0);
    run("v_mad_u64_u32", k_mad64, w, 0); run("v_cvt_f32_u32", k_cvt, w, 0); run("v_cvt_f32_u32 sdwa", k_cvtw, w, 0);
    run("v_mul_lo_u32", k_mullo, w, 0); run("v_mul_hi_u32", k_mulhi, w, 0); run("v_alignbit", k_align, w, 0);
    return 0;
}
