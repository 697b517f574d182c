// Philox4x32-10 (Salmon et al., SC'11), the counter-based generator behind every random draw of
// the render path.  It replaces the reference's global rand() (utils.h:25) and the per-pixel
// curandState of cuda/main.cu:281-291,340-348: a draw is a pure function of
//     counter = (pixel, sample, segment, 0)   key = (seed_lo, seed_hi)
// where segment 0 holds a sample's camera draws and segment k >= 1 the scatter draws after the
// k-th closest-hit query.  The block's words are the segment's first draws; a rejection loop
// that needs more steps the xoshiro128 engine from the block, once per attempt (rt_kernels.hip).  Uniforms are
// (word >> 8) * 2^-24, in [0,1) like random_double().
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rt {

// Device code only, same numbers whichever way:
// XOR3: the three-operand exclusive-or of a round as one instruction (inline assembly, which ties
// the register allocator's hands a little): 1 - 1.6 % of the frame for most kernels, five more
// spilled registers and 1 - 2 % lost for the one shortest of them (the walk out of HBM that also
// carries rect and texture code).
// ORDERED: see the barrier below.  The caller chooses both.
template <bool XOR3 = false, bool ORDERED = false>
__host__ __device__ __forceinline__ uint4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                        uint32_t k0, uint32_t k1) {
#if defined(__HIP_DEVICE_COMPILE__)
    // The key is the same in every lane, so the compiler works the ten round keys out once per launch
    // and keeps them in twenty scalar registers.  The render kernels have none to spare: the keys end
    // up parked in the lanes of a vector register, and every use is a v_readlane and a wait state.
    // Behind this barrier they are worked out where they are used instead (scalar adds, which cost the
    // vector pipes nothing): headline frame -1.9 %, Cornell -0.6 %, 10^6 spheres -1.4 %.  (Kept per
    // lane in vector registers they gained the sphere-only kernels 0.8 % and cost the kernels that
    // carry rect and texture code, short of vector registers, 1.6 %.)
    // The barrier comes in two strengths.  A `volatile` statement counts as a possible write to
    // memory: no load moves across it, and every load of a scene record after it goes through the
    // vector memory path instead of a scalar load.  That costs a room of turned boxes (twelve rect
    // and transform records per ray) 20 % and Cornell 2.4 %, and it GAINS the walk out of HBM 2.6 %
    // (10^6 spheres; the shading loads stay behind the block instead of being live across it).  The
    // plain statement with the counter as a dummy input stays inside the path loop all the same
    // (its result depends on the counter) and leaves the loads alone.
    if constexpr (ORDERED) asm volatile("" : "+s"(k0), "+s"(k1));
    else asm("" : "+s"(k0), "+s"(k1) : "v"(c0));
#endif
#pragma unroll
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
#if defined(__HIP_DEVICE_COMPILE__)
        if constexpr (XOR3) {
            // v_bitop3 with the truth table of a ^ b ^ c (the compiler emits two two-operand
            // instructions, the second with the scalar key at half rate)
            asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x96" : "=v"(n0) : "v"((uint32_t)(p1 >> 32)), "v"(c1), "s"(k0));
            asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x96" : "=v"(n2) : "v"((uint32_t)(p0 >> 32)), "v"(c3), "s"(k1));
        }
#endif
        c1 = (uint32_t)p1;
        c3 = (uint32_t)p0;
        c0 = n0;
        c2 = n2;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    uint4 out;
    out.x = c0; out.y = c1; out.z = c2; out.w = c3;
    return out;
}

}  // namespace rt
