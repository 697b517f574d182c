// Texture file decoding on the host: JPEG (baseline and progressive Huffman, 8 bit) -> RGB8 rows,
// top row first.  Takes the place of the stbi_load call in load_texture (cuda/main.cu:393; the
// reference vendors stb_image v2.27, cuda/stb_image.h).  The arithmetic that decides pixel
// values follows that decoder so that a texture holds the same texels as in the reference:
//   * inverse DCT: the LL&M "slow integer" factorisation with 12-bit constants, two extra
//     bits kept between the column and the row pass (cuda/stb_image.h:2392-2490);
//   * chroma upsampling: centred 3:1 triangle filter for 2x factors, nearest otherwise
//     (cuda/stb_image.h:3411-3473, 3592-3601), rows paired as in cuda/stb_image.h:3872-3887;
//   * YCbCr -> RGB in 20-bit fixed point with the constants rounded to 12 bits
//     (cuda/stb_image.h:3605-3629).
// Not handled (RT_ERR_UNSUPPORTED): 12-bit samples, arithmetic coding, lossless and
// hierarchical processes.
#include "../../include/rt.h"

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

namespace {

thread_local std::string g_error;

struct DecodeError {
    int code;
    const char* what;
};
[[noreturn]] void corrupt(const char* what) { throw DecodeError{RT_ERR_INVALID, what}; }
[[noreturn]] void unsupported(const char* what) { throw DecodeError{RT_ERR_UNSUPPORTED, what}; }

// zigzag position -> row-major position in the 8x8 block
const uint8_t kNatural[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                              41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                              30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// ---- Huffman tables: one 16-bit lookup per table -------------------------------------------------

struct Huffman {
    bool present = false;
    std::vector<uint8_t> len, sym;       // indexed by the next 16 bits of the stream; len 0 = no such code
    void build(const int counts[16], const uint8_t* symbols) {
        len.assign(65536, 0);
        sym.assign(65536, 0);
        unsigned code = 0;
        int k = 0;
        for (int l = 1; l <= 16; l++) {
            for (int i = 0; i < counts[l - 1]; i++, k++) {
                if (code >= (1u << l)) corrupt("bad code lengths");
                unsigned first = code << (16 - l), last = (code + 1) << (16 - l);
                for (unsigned e = first; e < last; e++) { len[e] = (uint8_t)l; sym[e] = symbols[k]; }
                code++;
            }
            code <<= 1;
        }
        present = true;
    }
};

// ---- entropy-coded segment reader ---------------------------------------------------------------------

struct BitReader {
    const uint8_t* p;
    const uint8_t* end;
    uint64_t acc = 0;
    int n = 0;               // valid bits at the bottom of acc
    int marker = -1;         // marker that ended the segment (its code), once met; zeros are fed after it

    void fill() {
        while (n <= 48) {
            unsigned b = 0;
            if (marker < 0 && p < end) {
                b = *p++;
                if (b == 0xff) {
                    unsigned c = p < end ? *p++ : 0;
                    while (c == 0xff && p < end) c = *p++;     // fill bytes
                    if (c != 0) { marker = (int)c; b = 0; }
                }
            }
            acc = (acc << 8) | b;
            n += 8;
        }
    }
    unsigned peek16() {
        if (n < 16) fill();
        return (unsigned)(acc >> (n - 16)) & 0xffffu;
    }
    int decode(const Huffman& h) {
        unsigned v = peek16();
        int l = h.len[v];
        if (l == 0) corrupt("bad huffman code");
        n -= l;
        return h.sym[v];
    }
    int bits(int count) {
        if (count == 0) return 0;
        if (n < count) fill();
        n -= count;
        return (int)((acc >> n) & ((1u << count) - 1u));
    }
    int bit() { return bits(1); }
    // `count` bits as a signed difference (JPEG "receive and extend")
    int receive_extend(int count) {
        int v = bits(count);
        return v < (1 << (count - 1)) ? v - (1 << count) + 1 : v;
    }
    void restart(const uint8_t* at) { p = at; acc = 0; n = 0; marker = -1; }
};

// ---- inverse DCT ------------------------------------------------------------------------------------------

constexpr int fx(float x) { return (int)((double)(x * 4096.0f) + 0.5); }      // 12-bit fixed point, rounded as the reference rounds

// Sums and products are formed in unsigned arithmetic (wrapping, the same bits as int where
// nothing overflows) so that the extreme coefficients of a damaged file stay defined behaviour.
typedef unsigned U;
struct Idct1D { U e0, e1, e2, e3, o0, o1, o2, o3; };   // even part sums, odd part terms: out[k] = e[k] +- o[3-k]

inline Idct1D idct_1d(int i0, int i1, int i2, int i3, int i4, int i5, int i6, int i7) {
    const U s0 = (U)i0, s1 = (U)i1, s2 = (U)i2, s3 = (U)i3, s4 = (U)i4, s5 = (U)i5, s6 = (U)i6, s7 = (U)i7;
    Idct1D r;
    U z = (s2 + s6) * (U)fx(0.5411961f);
    U t2 = z + s6 * (U)fx(-1.847759065f);
    U t3 = z + s2 * (U)fx(0.765366865f);
    U t0 = (s0 + s4) * 4096u, t1 = (s0 - s4) * 4096u;
    r.e0 = t0 + t3; r.e3 = t0 - t3; r.e1 = t1 + t2; r.e2 = t1 - t2;
    U a = s7, b = s5, c = s3, d = s1;
    U p3 = a + c, p4 = b + d, p1 = a + d, p2 = b + c;
    U p5 = (p3 + p4) * (U)fx(1.175875602f);
    a *= (U)fx(0.298631336f); b *= (U)fx(2.053119869f); c *= (U)fx(3.072711026f); d *= (U)fx(1.501321110f);
    p1 = p5 + p1 * (U)fx(-0.899976223f);
    p2 = p5 + p2 * (U)fx(-2.562915447f);
    p3 *= (U)fx(-1.961570560f);
    p4 *= (U)fx(-0.390180644f);
    r.o3 = d + p1 + p4; r.o2 = c + p2 + p3; r.o1 = b + p2 + p4; r.o0 = a + p1 + p3;
    return r;
}

inline uint8_t clamp8(int x) { return (uint8_t)(x < 0 ? 0 : (x > 255 ? 255 : x)); }
inline int sar(U x, int n) { return (int)x >> n; }           // arithmetic shift of the wrapped sum

void idct_block(uint8_t* out, int stride, const int16_t* d) {
    int v[64];
    for (int i = 0; i < 8; i++) {
        const int16_t* c = d + i;
        if (!c[8] && !c[16] && !c[24] && !c[32] && !c[40] && !c[48] && !c[56]) {
            int dc = c[0] * 4;
            for (int k = 0; k < 8; k++) v[8 * k + i] = dc;
            continue;
        }
        Idct1D r = idct_1d(c[0], c[8], c[16], c[24], c[32], c[40], c[48], c[56]);
        const U rnd = 512u;                                    // keep two extra bits: >> 10 instead of >> 12
        v[i] = sar(r.e0 + rnd + r.o3, 10);      v[56 + i] = sar(r.e0 + rnd - r.o3, 10);
        v[8 + i] = sar(r.e1 + rnd + r.o2, 10);  v[48 + i] = sar(r.e1 + rnd - r.o2, 10);
        v[16 + i] = sar(r.e2 + rnd + r.o1, 10); v[40 + i] = sar(r.e2 + rnd - r.o1, 10);
        v[24 + i] = sar(r.e3 + rnd + r.o0, 10); v[32 + i] = sar(r.e3 + rnd - r.o0, 10);
    }
    for (int i = 0; i < 8; i++, out += stride) {
        const int* w = v + 8 * i;
        Idct1D r = idct_1d(w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7]);
        const U rnd = 65536u + (128u << 17);                   // 2^-17 scale, rounded; level shift by 128
        out[0] = clamp8(sar(r.e0 + rnd + r.o3, 17)); out[7] = clamp8(sar(r.e0 + rnd - r.o3, 17));
        out[1] = clamp8(sar(r.e1 + rnd + r.o2, 17)); out[6] = clamp8(sar(r.e1 + rnd - r.o2, 17));
        out[2] = clamp8(sar(r.e2 + rnd + r.o1, 17)); out[5] = clamp8(sar(r.e2 + rnd - r.o1, 17));
        out[3] = clamp8(sar(r.e3 + rnd + r.o0, 17)); out[4] = clamp8(sar(r.e3 + rnd - r.o0, 17));
    }
}

// ---- decoder state ----------------------------------------------------------------------------------------

struct Component {
    int id = 0, h = 1, v = 1, tq = 0;
    int td = 0, ta = 0;                  // Huffman tables of the current scan
    int x = 0, y = 0;                    // samples this component really has
    int w2 = 0, h2 = 0;                  // plane size: whole MCUs
    int dc_pred = 0;
    std::vector<uint8_t> plane;
    std::vector<int16_t> coeff;          // progressive: 64 per block, blocks in rows of w2 / 8
};

struct Jpeg {
    const uint8_t* p;
    const uint8_t* end;
    int width = 0, height = 0, ncomp = 0;
    bool progressive = false, have_frame = false, jfif = false;
    int adobe_transform = -1, rgb_ids = 0;
    int restart_interval = 0;
    int hmax = 1, vmax = 1, mcu_x = 0, mcu_y = 0;
    uint16_t quant[4][64] = {};
    Huffman dc[4], ac[4];
    Component comp[4];
    // current scan
    int scan_n = 0, order[4] = {0, 0, 0, 0};
    int ss = 0, se = 63, ah = 0, al = 0;
    int eob_run = 0;
    int pending = -1;        // marker met while reading entropy-coded data, not processed yet

    unsigned u8() { if (p >= end) corrupt("truncated file"); return *p++; }
    unsigned u16() { unsigned a = u8(); return (a << 8) | u8(); }
    void skip(long n) { if (n < 0 || n > end - p) corrupt("truncated file"); p += n; }

    int next_marker() {
        // a marker is 0xff (possibly repeated) and a code; bytes in between segments are skipped
        for (;;) {
            if (p >= end) corrupt("no marker where one is expected");
            if (*p++ != 0xff) continue;
            while (p < end && *p == 0xff) p++;
            if (p >= end) corrupt("truncated file");
            unsigned c = *p++;
            if (c != 0) return (int)c;
        }
    }

    void read_dqt() {
        long L = (long)u16() - 2;
        while (L > 0) {
            unsigned q = u8();
            int prec = q >> 4, t = q & 15;
            if (prec > 1) corrupt("bad DQT type");
            if (t > 3) corrupt("bad DQT table");
            for (int i = 0; i < 64; i++) quant[t][kNatural[i]] = (uint16_t)(prec ? u16() : u8());
            L -= prec ? 129 : 65;
        }
        if (L != 0) corrupt("bad DQT length");
    }

    void read_dht() {
        long L = (long)u16() - 2;
        while (L > 0) {
            unsigned q = u8();
            int tc = q >> 4, th = q & 15;
            if (tc > 1 || th > 3) corrupt("bad DHT header");
            int counts[16], n = 0;
            for (int i = 0; i < 16; i++) { counts[i] = (int)u8(); n += counts[i]; }
            if (n > 256) corrupt("bad DHT counts");
            uint8_t symbols[256];
            for (int i = 0; i < n; i++) symbols[i] = (uint8_t)u8();
            (tc == 0 ? dc[th] : ac[th]).build(counts, symbols);
            L -= 17 + n;
        }
        if (L != 0) corrupt("bad DHT length");
    }

    void read_app(int m) {
        long L = (long)u16();
        if (L < 2) corrupt("bad segment length");
        L -= 2;
        if (m == 0xe0 && L >= 5) {
            static const uint8_t tag[5] = {'J', 'F', 'I', 'F', 0};
            bool ok = true;
            for (int i = 0; i < 5; i++) ok = (u8() == tag[i]) && ok;
            L -= 5;
            if (ok) jfif = true;
        } else if (m == 0xee && L >= 12) {
            static const uint8_t tag[6] = {'A', 'd', 'o', 'b', 'e', 0};
            bool ok = true;
            for (int i = 0; i < 6; i++) ok = (u8() == tag[i]) && ok;
            L -= 6;
            if (ok) { u8(); u16(); u16(); adobe_transform = (int)u8(); L -= 6; }
        }
        skip(L);
    }

    void read_frame(int m) {
        if (have_frame) corrupt("second frame header");
        progressive = m == 0xc2;
        unsigned Lf = u16();
        if (Lf < 11) corrupt("bad SOF length");
        if (u8() != 8) unsupported("only 8-bit samples");
        height = (int)u16();
        width = (int)u16();
        if (height == 0) unsupported("height given by a DNL segment");
        if (width == 0) corrupt("zero width");
        ncomp = (int)u8();
        if (ncomp != 1 && ncomp != 3 && ncomp != 4) corrupt("bad component count");
        if (Lf != 8u + 3u * (unsigned)ncomp) corrupt("bad SOF length");
        for (int i = 0; i < ncomp; i++) {
            Component& c = comp[i];
            c.id = (int)u8();
            if (ncomp == 3 && c.id == "RGB"[i]) rgb_ids++;
            unsigned q = u8();
            c.h = q >> 4; c.v = q & 15;
            if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4) corrupt("bad sampling factor");
            c.tq = (int)u8();
            if (c.tq > 3) corrupt("bad quantisation table index");
            if (c.h > hmax) hmax = c.h;
            if (c.v > vmax) vmax = c.v;
        }
        for (int i = 0; i < ncomp; i++)
            if (hmax % comp[i].h || vmax % comp[i].v) unsupported("fractional sampling ratio");
        if ((uint64_t)width * (uint64_t)height > (1ull << 28)) unsupported("image larger than 2^28 pixels");
        mcu_x = (width + 8 * hmax - 1) / (8 * hmax);
        mcu_y = (height + 8 * vmax - 1) / (8 * vmax);
        for (int i = 0; i < ncomp; i++) {
            Component& c = comp[i];
            c.x = (width * c.h + hmax - 1) / hmax;
            c.y = (height * c.v + vmax - 1) / vmax;
            c.w2 = mcu_x * c.h * 8;
            c.h2 = mcu_y * c.v * 8;
            c.plane.assign((size_t)c.w2 * c.h2, 0);
            if (progressive) c.coeff.assign((size_t)c.w2 * c.h2, 0);
        }
        have_frame = true;
    }

    void read_scan_header() {
        if (!have_frame) corrupt("scan before frame header");
        unsigned Ls = u16();
        scan_n = (int)u8();
        if (scan_n < 1 || scan_n > 4 || scan_n > ncomp) corrupt("bad SOS component count");
        if (Ls != 6u + 2u * (unsigned)scan_n) corrupt("bad SOS length");
        for (int i = 0; i < scan_n; i++) {
            int id = (int)u8();
            unsigned q = u8();
            int which = 0;
            while (which < ncomp && comp[which].id != id) which++;
            if (which == ncomp) corrupt("scan names an unknown component");
            comp[which].td = q >> 4; comp[which].ta = q & 15;
            if (comp[which].td > 3 || comp[which].ta > 3) corrupt("bad Huffman table index");
            order[i] = which;
        }
        ss = (int)u8(); se = (int)u8();
        unsigned a = u8();
        ah = a >> 4; al = a & 15;
        if (progressive) {
            if (ss > 63 || se > 63 || ss > se || ah > 13 || al > 13) corrupt("bad SOS");
        } else {
            if (ss != 0 || ah != 0 || al != 0) corrupt("bad SOS");
            se = 63;
        }
    }

    // ---- blocks ------------------------------------------------------------------------------------------

    static const Huffman& need(const Huffman& h) { if (!h.present) corrupt("scan uses an undefined Huffman table"); return h; }

    void block_sequential(BitReader& br, Component& c, int16_t* out) {
        const Huffman& hd = need(dc[c.td]);
        const Huffman& ha = need(ac[c.ta]);
        const uint16_t* q = quant[c.tq];
        std::memset(out, 0, 64 * sizeof(int16_t));
        int t = br.decode(hd);
        if (t > 15) corrupt("bad DC size");
        c.dc_pred = (int)((U)c.dc_pred + (U)(t ? br.receive_extend(t) : 0));
        out[0] = (int16_t)((U)c.dc_pred * q[0]);
        for (int k = 1; k < 64;) {
            int rs = br.decode(ha), s = rs & 15, r = rs >> 4;
            if (s == 0) {
                if (rs != 0xf0) break;
                k += 16;
                continue;
            }
            k += r;
            if (k > 63) corrupt("coefficient index past the block");
            int z = kNatural[k++];
            out[z] = (int16_t)(br.receive_extend(s) * q[z]);
        }
    }

    void block_dc_progressive(BitReader& br, Component& c, int16_t* blk) {
        if (se != 0) corrupt("DC scan with AC coefficients");
        if (ah == 0) {
            std::memset(blk, 0, 64 * sizeof(int16_t));
            int t = br.decode(need(dc[c.td]));
            if (t > 15) corrupt("bad DC size");
            c.dc_pred = (int)((U)c.dc_pred + (U)(t ? br.receive_extend(t) : 0));
            blk[0] = (int16_t)((U)c.dc_pred << al);
        } else if (br.bit()) {
            blk[0] = (int16_t)(blk[0] + (1 << al));
        }
    }

    // one more bit of an already non-zero coefficient
    static void refine(BitReader& br, int16_t* c, int bitval) {
        if (br.bit() && (*c & bitval) == 0) *c = (int16_t)(*c > 0 ? *c + bitval : *c - bitval);
    }

    void block_ac_progressive(BitReader& br, Component& c, int16_t* blk) {
        if (ss == 0) corrupt("AC scan starting at the DC coefficient");
        const Huffman& ha = need(ac[c.ta]);
        if (ah == 0) {                                       // first pass over this band
            if (eob_run) { eob_run--; return; }
            for (int k = ss; k <= se;) {
                int rs = br.decode(ha), s = rs & 15, r = rs >> 4;
                if (s == 0) {
                    if (r < 15) {                            // end of band for 2^r + extra blocks, this one included
                        eob_run = (1 << r) + br.bits(r) - 1;
                        break;
                    }
                    k += 16;
                    continue;
                }
                k += r;
                if (k > 63) corrupt("coefficient index past the block");
                blk[kNatural[k++]] = (int16_t)(br.receive_extend(s) * (1 << al));
            }
            return;
        }
        const int bitval = 1 << al;                          // refinement pass
        if (eob_run) {
            eob_run--;
            for (int k = ss; k <= se; k++) {
                int16_t* p = &blk[kNatural[k]];
                if (*p) refine(br, p, bitval);
            }
            return;
        }
        for (int k = ss; k <= se;) {
            int rs = br.decode(ha), s = rs & 15, r = rs >> 4;
            int newval = 0;
            if (s == 0) {
                if (r < 15) {
                    eob_run = (1 << r) - 1 + br.bits(r);
                    r = 64;                                  // run to the end of the band, placing nothing
                }
            } else {
                if (s != 1) corrupt("bad refinement code");
                newval = br.bit() ? bitval : -bitval;
            }
            // skip r zero coefficients (refining the non-zero ones passed on the way), then place newval
            while (k <= se) {
                int16_t* p = &blk[kNatural[k++]];
                if (*p) { refine(br, p, bitval); continue; }
                if (r == 0) { *p = (int16_t)newval; break; }
                r--;
            }
        }
    }

    void decode_block(BitReader& br, Component& c, int bx, int by) {
        if (!progressive) {
            int16_t blk[64];
            block_sequential(br, c, blk);
            idct_block(c.plane.data() + (size_t)c.w2 * by * 8 + bx * 8, c.w2, blk);
        } else {
            int16_t* blk = c.coeff.data() + 64 * ((size_t)bx + (size_t)by * (c.w2 / 8));
            if (ss == 0) block_dc_progressive(br, c, blk);
            else block_ac_progressive(br, c, blk);
        }
    }

    // Counts one MCU off the restart interval.  Returns false when the scan has to stop: the
    // interval ran out and no restart marker follows.
    bool restart_point(BitReader& br, int& todo) {
        if (--todo > 0) return true;
        br.n = 0; br.acc = 0;                                // padding bits of the last byte
        br.fill();
        if (br.marker < 0xd0 || br.marker > 0xd7) return false;
        br.restart(br.p);
        for (int i = 0; i < 4; i++) comp[i].dc_pred = 0;
        eob_run = 0;
        todo = restart_interval ? restart_interval : 0x7fffffff;
        return true;
    }

    void decode_scan() {
        BitReader br;
        br.p = p; br.end = end;
        for (int i = 0; i < 4; i++) comp[i].dc_pred = 0;
        eob_run = 0;
        int todo = restart_interval ? restart_interval : 0x7fffffff;
        bool go = true;
        if (scan_n == 1) {
            Component& c = comp[order[0]];
            int bw = (c.x + 7) >> 3, bh = (c.y + 7) >> 3;   // only the blocks that hold real samples
            for (int by = 0; by < bh && go; by++)
                for (int bx = 0; bx < bw && go; bx++) {
                    decode_block(br, c, bx, by);
                    go = restart_point(br, todo);
                }
        } else {
            if (progressive && ss != 0) corrupt("interleaved AC scan");
            for (int my = 0; my < mcu_y && go; my++)
                for (int mx = 0; mx < mcu_x && go; mx++) {
                    for (int k = 0; k < scan_n; k++) {
                        Component& c = comp[order[k]];
                        for (int y = 0; y < c.v; y++)
                            for (int x = 0; x < c.h; x++) decode_block(br, c, mx * c.h + x, my * c.v + y);
                    }
                    go = restart_point(br, todo);
                }
        }
        // go on at the marker that ended the entropy-coded data, or look for the next one
        p = br.p;
        if (br.marker >= 0) pending = br.marker;
    }

    void finish_progressive() {
        for (int n = 0; n < ncomp; n++) {
            Component& c = comp[n];
            int bw = (c.x + 7) >> 3, bh = (c.y + 7) >> 3;
            for (int by = 0; by < bh; by++)
                for (int bx = 0; bx < bw; bx++) {
                    int16_t* blk = c.coeff.data() + 64 * ((size_t)bx + (size_t)by * (c.w2 / 8));
                    for (int i = 0; i < 64; i++) blk[i] = (int16_t)(blk[i] * quant[c.tq][i]);
                    idct_block(c.plane.data() + (size_t)c.w2 * by * 8 + bx * 8, c.w2, blk);
                }
        }
    }

    void decode() {
        if (end - p < 2 || p[0] != 0xff || p[1] != 0xd8) corrupt("not a JPEG file");
        p += 2;
        for (;;) {
            int m = pending >= 0 ? pending : next_marker();
            pending = -1;
            if (m == 0xd9) break;
            if (m == 0xdb) read_dqt();
            else if (m == 0xc4) read_dht();
            else if (m == 0xdd) { if (u16() != 4) corrupt("bad DRI length"); restart_interval = (int)u16(); }
            else if ((m >= 0xe0 && m <= 0xef) || m == 0xfe) read_app(m);
            else if (m == 0xc0 || m == 0xc1 || m == 0xc2) read_frame(m);
            else if (m == 0xda) { read_scan_header(); decode_scan(); }
            else if (m == 0xdc) { if (u16() != 4) corrupt("bad DNL length"); if ((int)u16() != height) corrupt("bad DNL height"); }
            else if (m == 0xc3 || (m >= 0xc5 && m <= 0xcf)) unsupported("JPEG process other than baseline or progressive Huffman");
            else if (m >= 0xd0 && m <= 0xd7) continue;      // stray restart marker
            else corrupt("unknown marker");
        }
        if (!have_frame) corrupt("no frame header");
        if (progressive) finish_progressive();
    }
};

// ---- upsampling and colour ------------------------------------------------------------------------------------

inline uint8_t div4(int x) { return (uint8_t)(x >> 2); }
inline uint8_t div16(int x) { return (uint8_t)(x >> 4); }

// One output row of a component: `near` is the source row this output row lies in, `far` the
// neighbouring source row on the side it leans to.  Returns where the row is (near itself when
// nothing has to be computed).
const uint8_t* upsample_row(uint8_t* out, const uint8_t* near, const uint8_t* far, int w, int hs, int vs) {
    if (hs == 1 && vs == 1) return near;
    if (hs == 1 && vs == 2) {
        for (int i = 0; i < w; i++) out[i] = div4(3 * near[i] + far[i] + 2);
        return out;
    }
    if (hs == 2 && vs == 1) {
        if (w == 1) { out[0] = out[1] = near[0]; return out; }
        out[0] = near[0];
        out[1] = div4(near[0] * 3 + near[1] + 2);
        int i;
        for (i = 1; i < w - 1; i++) {
            int n = 3 * near[i] + 2;
            out[2 * i] = div4(n + near[i - 1]);
            out[2 * i + 1] = div4(n + near[i + 1]);
        }
        out[2 * i] = div4(near[w - 2] * 3 + near[w - 1] + 2);
        out[2 * i + 1] = near[w - 1];
        return out;
    }
    if (hs == 2 && vs == 2) {
        if (w == 1) { out[0] = out[1] = div4(3 * near[0] + far[0] + 2); return out; }
        int t1 = 3 * near[0] + far[0];
        out[0] = div4(t1 + 2);
        for (int i = 1; i < w; i++) {
            int t0 = t1;
            t1 = 3 * near[i] + far[i];
            out[2 * i - 1] = div16(3 * t0 + t1 + 8);
            out[2 * i] = div16(3 * t1 + t0 + 8);
        }
        out[2 * w - 1] = div4(t1 + 2);
        return out;
    }
    for (int i = 0; i < w; i++)
        for (int j = 0; j < hs; j++) out[i * hs + j] = near[i];
    return out;
}

constexpr int cfix(float x) { return ((int)(x * 4096.0f + 0.5f)) << 8; }

inline void ycc_to_rgb(uint8_t* out, int y, int cb, int cr) {
    int yf = (y << 20) + (1 << 19);
    cr -= 128; cb -= 128;
    int r = yf + cr * cfix(1.40200f);
    int g = yf + cr * -cfix(0.71414f) + (int)((unsigned)(cb * -cfix(0.34414f)) & 0xffff0000u);
    int b = yf + cb * cfix(1.77200f);
    out[0] = clamp8(r >> 20); out[1] = clamp8(g >> 20); out[2] = clamp8(b >> 20);
}

inline uint8_t mul8(uint8_t x, uint8_t y) {                 // x * y / 255, rounded
    unsigned t = (unsigned)x * y + 128;
    return (uint8_t)((t + (t >> 8)) >> 8);
}

void to_rgb(Jpeg& j, uint8_t* rgb) {
    const int W = j.width, H = j.height, n = j.ncomp;
    const bool is_rgb = n == 3 && (j.rgb_ids == 3 || (j.adobe_transform == 0 && !j.jfif));
    std::vector<uint8_t> line[4];
    const uint8_t* row[4] = {nullptr, nullptr, nullptr, nullptr};
    for (int k = 0; k < n; k++) line[k].resize((size_t)W + 8);
    for (int y = 0; y < H; y++) {
        for (int k = 0; k < n; k++) {
            Component& c = j.comp[k];
            int hs = j.hmax / c.h, vs = j.vmax / c.v;
            // output row y lies in source row y / vs; with a factor of two it leans to the row
            // above in its upper half and to the row below in its lower half
            int near = y / vs, far = near;
            if (vs == 2) far = (y & 1) ? near + 1 : near - 1;
            if (far < 0) far = 0;
            if (far > c.y - 1) far = c.y - 1;
            if (near > c.y - 1) near = c.y - 1;
            int w_lores = (W + hs - 1) / hs;
            row[k] = upsample_row(line[k].data(), c.plane.data() + (size_t)c.w2 * near, c.plane.data() + (size_t)c.w2 * far,
                                  w_lores, hs, vs);
        }
        uint8_t* out = rgb + (size_t)3 * W * y;
        for (int i = 0; i < W; i++, out += 3) {
            if (n == 1) { out[0] = out[1] = out[2] = row[0][i]; continue; }
            if (n == 3) {
                if (is_rgb) { out[0] = row[0][i]; out[1] = row[1][i]; out[2] = row[2][i]; }
                else ycc_to_rgb(out, row[0][i], row[1][i], row[2][i]);
                continue;
            }
            uint8_t m = row[3][i];
            if (j.adobe_transform == 0) {                        // CMYK
                out[0] = mul8(row[0][i], m); out[1] = mul8(row[1][i], m); out[2] = mul8(row[2][i], m);
            } else {
                ycc_to_rgb(out, row[0][i], row[1][i], row[2][i]);
                if (j.adobe_transform == 2) {                    // YCCK
                    out[0] = mul8((uint8_t)(255 - out[0]), m); out[1] = mul8((uint8_t)(255 - out[1]), m); out[2] = mul8((uint8_t)(255 - out[2]), m);
                }
            }
        }
    }
}

}  // namespace

extern "C" {

int rt_decode_jpeg(const uint8_t* bytes, uint64_t n_bytes, int32_t* width, int32_t* height, uint8_t** rgb8) {
    if (!bytes || !width || !height || !rgb8) { g_error = "null argument"; return RT_ERR_INVALID; }
    *rgb8 = nullptr; *width = 0; *height = 0;
    try {
        std::unique_ptr<Jpeg> j(new Jpeg);
        j->p = bytes; j->end = bytes + n_bytes;
        j->decode();
        uint8_t* out = (uint8_t*)std::malloc((size_t)3 * j->width * j->height);
        if (!out) { g_error = "out of memory"; return RT_ERR_INVALID; }
        to_rgb(*j, out);
        *rgb8 = out; *width = j->width; *height = j->height;
        return RT_OK;
    } catch (const DecodeError& e) {
        g_error = e.what;
        return e.code;
    } catch (const std::bad_alloc&) {
        g_error = "out of memory";
        return RT_ERR_INVALID;
    }
}

int rt_load_jpeg_file(const char* path, int32_t* width, int32_t* height, uint8_t** rgb8) {
    if (!path) { g_error = "null path"; return RT_ERR_INVALID; }
    FILE* f = std::fopen(path, "rb");
    if (!f) { g_error = std::string("cannot open ") + path; return RT_ERR_INVALID; }
    std::vector<uint8_t> buf;
    uint8_t chunk[65536];
    size_t got;
    while ((got = std::fread(chunk, 1, sizeof chunk, f)) > 0) buf.insert(buf.end(), chunk, chunk + got);
    std::fclose(f);
    return rt_decode_jpeg(buf.data(), buf.size(), width, height, rgb8);
}

void rt_free_image(uint8_t* rgb8) { std::free(rgb8); }

// cuda/main.cu:358-383 / main.cc:109-126: "P3", size, 255, then one pixel per line, top row first.
// The decimal text of a byte comes from a table: a 1080p frame takes 5 ms of host time that way (1.5 ms
// as binary P6); formatting its six million numbers through printf took 150 ms, several frames' worth
// of rendering.
int rt_write_ppm(const char* path, const uint8_t* rgb8, int32_t width, int32_t height, int32_t binary) {
    if (!path || !rgb8 || width < 1 || height < 1) { g_error = "rt_write_ppm: null path or pixels, or an empty image"; return RT_ERR_INVALID; }
    const size_t w = (size_t)width, h = (size_t)height;
    char head[64];
    const int head_len = std::snprintf(head, sizeof head, "P%d\n%d %d\n255\n", binary ? 6 : 3, width, height);
    std::vector<char> out;
    if (binary) {
        out.resize((size_t)head_len + 3 * w * h);
        std::memcpy(out.data(), head, (size_t)head_len);
        for (size_t j = 0; j < h; j++) std::memcpy(out.data() + head_len + 3 * w * j, rgb8 + 3 * w * (h - 1 - j), 3 * w);
    } else {
        char digits[256][4];
        int len[256];
        for (int v = 0; v < 256; v++) len[v] = std::snprintf(digits[v], 4, "%d", v);
        out.resize((size_t)head_len + 12 * w * h);             // at most three digits and a separator per number
        char* q = out.data();
        std::memcpy(q, head, (size_t)head_len); q += head_len;
        for (size_t j = h; j-- > 0;)
            for (size_t i = 0; i < w; i++) {
                const uint8_t* p = rgb8 + 3 * (j * w + i);
                for (int c = 0; c < 3; c++) {
                    std::memcpy(q, digits[p[c]], 4); q += len[p[c]];    // (what follows the digits is overwritten next)
                    *q++ = c < 2 ? ' ' : '\n';
                }
            }
        out.resize((size_t)(q - out.data()));
    }
    FILE* f = std::fopen(path, "wb");
    if (!f) { g_error = std::string("cannot open ") + path + " for writing"; return RT_ERR_INVALID; }
    const size_t put = std::fwrite(out.data(), 1, out.size(), f);
    const int closed = std::fclose(f);
    if (put != out.size() || closed != 0) { g_error = std::string("short write to ") + path; return RT_ERR_INVALID; }
    return RT_OK;
}

const char* rt_image_last_error(void) { return g_error.c_str(); }

}  // extern "C"
