/* TEST INFRASTRUCTURE — CPU checker only; nothing shipped links this file.
 *
 * Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy
 * as 1, 2, 3", SC'11) and the draw layout of the keyed render stream.
 *
 * The reference draws every random number through rand() (utils.h:25,
 * random_double = rand()/(RAND_MAX+1.0) in [0,1)).  The keyed stream replaces
 * that global sequence by a pure function of
 *     (seed, pixel, sample, segment, draw index inside the segment)
 * where segment 0 holds the camera draws of a sample (main.cc:117-119,
 * camera.h:41) and segment k >= 1 holds the scatter draws made after the k-th
 * closest-hit query of the path (main.cc:18,22).
 *
 * Draw d of a segment reads one 32-bit word of a Philox block:
 *     segment 0 : block d/4, word d%4   (two jitters, then disk pairs)
 *     segment>=1: block d/3, word d%3   (sphere rejection triples; word 3 unused)
 * counter = {pixel, sample, segment, block}, key = {seed_lo, seed_hi}.
 * A uniform is (word >> 8) * 2^-24: exactly representable in float and double.
 */
#ifndef RT_ORACLE_PHILOX_H
#define RT_ORACLE_PHILOX_H

#include <stdint.h>

static inline void rto_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2],
                                     uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* 24-bit integer behind draw d of (pixel, sample, segment). */
static inline uint32_t rto_draw_k24(uint64_t seed, uint32_t pixel, uint32_t sample,
                                    uint32_t segment, uint32_t d) {
    uint32_t block, word;
    if (segment == 0) { block = d >> 2; word = d & 3u; }
    else              { block = d / 3u; word = d % 3u; }
    uint32_t ctr[4] = { pixel, sample, segment, block };
    uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
    uint32_t out[4];
    rto_philox4x32_10(ctr, key, out);
    return out[word] >> 8;
}

#endif
