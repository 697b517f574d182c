/* TEST INFRASTRUCTURE — CPU checker only; nothing shipped links this file.
 *
 * Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy
 * as 1, 2, 3", SC'11), xoshiro128++ (Blackman, Vigna 2019) and the draw layout of the
 * keyed render stream.
 *
 * The reference draws every random number through rand() (utils.h:25,
 * random_double = rand()/(RAND_MAX+1.0) in [0,1)).  The keyed stream replaces
 * that global sequence by a pure function of
 *     (seed, pixel, sample, segment, draw index inside the segment)
 * where segment 0 holds the camera draws of a sample (main.cc:117-119,
 * camera.h:41) and segment k >= 1 holds the scatter draws made after the k-th
 * closest-hit query of the path (main.cc:18,22).
 *
 * One Philox block per segment: (w0,w1,w2,w3) = Philox(counter = {pixel, sample,
 * segment, 0}, key = {seed_lo, seed_hi}).  The first draws of the segment are
 * those words themselves (four in segment 0: two jitters and the first disk
 * pair; three in a scatter segment: the first sphere rejection triple); later
 * draws, needed only when a rejection attempt fails, are successive outputs of
 * xoshiro128++ started from the state (w0,w1,w2,w3).
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

typedef struct {
    uint32_t w[4];      /* the segment's Philox block */
    uint32_t s[4];      /* xoshiro128++ state */
    uint32_t d;         /* draws served so far */
    uint32_t n_direct;  /* draws served straight from w[] */
} rto_stream;

static inline uint32_t rto_rotl(uint32_t x, int k) { return (x << k) | (x >> (32 - k)); }

static inline uint32_t rto_xoshiro128pp(uint32_t s[4]) {
    uint32_t result = rto_rotl(s[0] + s[3], 7) + s[0];
    uint32_t t = s[1] << 9;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
    s[2] ^= t;
    s[3] = rto_rotl(s[3], 11);
    return result;
}

static inline void rto_stream_begin(rto_stream* st, uint64_t seed, uint32_t pixel, uint32_t sample,
                                    uint32_t segment) {
    uint32_t ctr[4] = { pixel, sample, segment, 0 };
    uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
    rto_philox4x32_10(ctr, key, st->w);
    for (int i = 0; i < 4; i++) st->s[i] = st->w[i];
    st->d = 0;
    st->n_direct = segment == 0 ? 4u : 3u;
}

/* next 24-bit integer of the segment */
static inline uint32_t rto_stream_next(rto_stream* st) {
    uint32_t word = st->d < st->n_direct ? st->w[st->d] : rto_xoshiro128pp(st->s);
    st->d++;
    return word >> 8;
}

/* 24-bit integer behind draw d of (pixel, sample, segment), stateless form. */
static inline uint32_t rto_draw_k24(uint64_t seed, uint32_t pixel, uint32_t sample,
                                    uint32_t segment, uint32_t d) {
    rto_stream st;
    rto_stream_begin(&st, seed, pixel, sample, segment);
    uint32_t k = 0;
    for (uint32_t i = 0; i <= d; i++) k = rto_stream_next(&st);
    return k;
}

#endif
