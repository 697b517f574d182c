/* TEST INFRASTRUCTURE — CPU checker only; nothing shipped links this file.
 *
 * Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy
 * as 1, 2, 3", SC'11), the xoshiro128 engine (Blackman, Vigna 2019) and the draw layout of
 * the keyed render stream.
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
 * pair; three in a scatter segment: the first sphere rejection triple).  Later
 * draws are needed only when a rejection attempt fails, and come one ATTEMPT per
 * step of the xoshiro128 linear engine started from the state (w0,w1,w2,w3): the
 * engine is stepped twice before the first of them (so that every state word has
 * mixed in w3, which no direct draw of a scatter segment uses) and once per further
 * attempt, and the attempt's draws are s0 + s3, s1 + s2 and (third draw of a sphere
 * triple) s0 - s3 of the new state.  Only bits 8..31 of a word are used: for a
 * uniformly distributed state those parts of the three words are exactly uniform
 * and independent (the first two use disjoint state words; s0 - s3 = (s0 + s3) -
 * 2 s3 shares only its lowest bit with s0 + s3), and the sum is the scrambler of
 * xoshiro128+, whose weak bits are the lowest ones.  One engine step per attempt
 * instead of one generator call per draw is what the rejection loops of the GPU
 * path cost: a wavefront iterates them until its unluckiest lane is done.
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
    uint32_t s[4];      /* xoshiro128 state */
    uint32_t d;         /* draws served so far */
    uint32_t n_direct;  /* draws served straight from w[] */
    uint32_t group;     /* draws per rejection attempt: 2 in segment 0 (lens disk), 3 in a scatter segment (sphere) */
} rto_stream;

static inline uint32_t rto_rotl(uint32_t x, int k) { return (x << k) | (x >> (32 - k)); }

/* one step of the xoshiro128 linear engine */
static inline void rto_xoshiro128_step(uint32_t s[4]) {
    uint32_t t = s[1] << 9;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
    s[2] ^= t;
    s[3] = rto_rotl(s[3], 11);
}

static inline void rto_stream_begin(rto_stream* st, uint64_t seed, uint32_t pixel, uint32_t sample,
                                    uint32_t segment) {
    uint32_t ctr[4] = { pixel, sample, segment, 0 };
    uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
    rto_philox4x32_10(ctr, key, st->w);
    for (int i = 0; i < 4; i++) st->s[i] = st->w[i];
    st->d = 0;
    st->n_direct = segment == 0 ? 4u : 3u;
    st->group = segment == 0 ? 2u : 3u;
}

/* next 24-bit integer of the segment */
static inline uint32_t rto_stream_next(rto_stream* st) {
    uint32_t word;
    if (st->d < st->n_direct) {
        word = st->w[st->d];
    } else {
        uint32_t j = (st->d - st->n_direct) % st->group;     /* position inside the attempt */
        if (j == 0) {
            if (st->d == st->n_direct) rto_xoshiro128_step(st->s);
            rto_xoshiro128_step(st->s);
        }
        word = j == 0 ? st->s[0] + st->s[3] : (j == 1 ? st->s[1] + st->s[2] : st->s[0] - st->s[3]);
    }
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
