// image_decode.cpp -- load_image_from_file for the drop-in API (replaces /root/reference/vit.cpp:109-127).
//
// The reference calls stbi_load(fname, &nx, &ny, &nc, 3) from stb_image.h, which lives in the absent ggml submodule
// (vit.h:5 includes "ggml/examples/stb_image.h"), so the decoder is written here from the file-format specifications:
//   * JPEG (ITU T.81): baseline / extended-sequential and PROGRESSIVE Huffman, 8-bit, 1, 3 or 4 components (YCbCr, Adobe RGB, CMYK and
//     YCCK through the APP14 colour-transform flag, as stbi_load(..., 3) converts them), sampling factors 1-2, restart intervals --
//     three of the reference's ten bundled images are progressive;
//   * PNG (RFC 2083) with its own zlib inflate: plain and Adam7-interlaced, colour types 0/2/3/4/6, bit depths 1-16, alpha dropped;
//   * binary PPM (P6), kept for the C++ example.
// Output is what stbi_load(..., 3) returns: tightly packed RGB u8, top row first.
// Where the standard leaves arithmetic to the decoder (IDCT, chroma upsampling, YCbCr -> RGB) the choices follow stb_image's
// published method: 12-bit fixed-point "islow" IDCT, 3:1 linear chroma interpolation, 20-bit fixed-point colour conversion.
// UNPINNED against stb itself (not in the tree); tests compare with PIL/libjpeg-turbo decodes of the bundled images within a
// stated +-LSB band (two correct JPEG decoders differ by rounding) and bit-exactly for PNG/PPM.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "model_file.h"

namespace {

struct DecodeError { std::string msg; };
[[noreturn]] void fail(const char *m) { throw DecodeError{m}; }

// ------------------------------------------------------------------------------------------------------------------
// JPEG
// ------------------------------------------------------------------------------------------------------------------
const uint8_t kZigzag[64 + 15] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
                                  35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
                                  63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};      // padding: corrupt run lengths cannot index out of range

struct Huff {
    bool present = false;
    uint8_t size[257];          // code length of the k-th symbol
    uint16_t code[256];
    uint8_t value[256];
    int maxcode[18];            // (largest code of length l) + 1, left-aligned to 16 bits
    int delta[17];              // symbol index = code + delta[l]
    int16_t fast[512];          // 9-bit prefix -> symbol index, or -1

    void build(const uint8_t counts[16], const uint8_t *vals, int nvals) {
        int k = 0;
        for (int l = 1; l <= 16; ++l)
            for (int i = 0; i < counts[l - 1]; ++i) { if (k >= 256) fail("jpeg: bad huffman table"); size[k++] = (uint8_t)l; }
        if (k != nvals) fail("jpeg: bad huffman table");
        size[k] = 0;
        int c = 0; k = 0;
        for (int l = 1; l <= 16; ++l) {
            delta[l] = k - c;
            if (size[k] == l) {
                while (size[k] == l) code[k++] = (uint16_t)c++;
                if (c - 1 >= (1 << l)) fail("jpeg: bad code lengths");
            }
            maxcode[l] = c << (16 - l);
            c <<= 1;
        }
        maxcode[17] = 0x7fffffff;
        memcpy(value, vals, nvals);
        for (int i = 0; i < 512; ++i) fast[i] = -1;
        for (int i = 0; i < nvals; ++i) {
            const int s = size[i];
            if (s <= 9) {
                const int c0 = code[i] << (9 - s), m = 1 << (9 - s);
                for (int j = 0; j < m; ++j) fast[c0 + j] = (int16_t)i;
            }
        }
        present = true;
    }
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int bw = 0, bh = 0;         // blocks allocated (whole MCUs)
    int pw = 0, ph = 0;         // pixels allocated
    int nbx = 0, nby = 0;       // blocks that hold image data (non-interleaved scans walk these)
    int dc_pred = 0;
    std::vector<int16_t> coef;  // [bh][bw][64], natural order, NOT dequantised
    std::vector<uint8_t> pix;   // [ph][pw]
};

struct Jpeg {
    const uint8_t *p, *end;
    int width = 0, height = 0, ncomp = 0, hmax = 1, vmax = 1;
    bool progressive = false;
    uint16_t qt[4][64]; bool qt_present[4] = {false, false, false, false};
    Huff hdc[4], hac[4];
    Component comp[4];
    int app14_transform = -1; bool jfif = false;      // Adobe APP14 colour transform (0 RGB/CMYK, 1 YCbCr, 2 YCCK; -1 = no marker), JFIF APP0 seen
    int restart_interval = 0;
    // bit reader
    uint32_t bitbuf = 0; int bitcnt = 0; int marker = -1; bool no_more = false;
    // scan
    int ss = 0, se = 63, ah = 0, al = 0, eobrun = 0;

    int u8() { if (p >= end) fail("jpeg: truncated"); return *p++; }
    int u16() { const int a = u8(); return (a << 8) | u8(); }

    void fill() {
        while (bitcnt <= 24) {
            int b = 0;
            if (!no_more) {
                if (p >= end) { no_more = true; }
                else {
                    b = *p++;
                    if (b == 0xFF) {
                        int c = p < end ? *p++ : 0xD9;
                        while (c == 0xFF) c = p < end ? *p++ : 0xD9;
                        if (c != 0) { marker = c; no_more = true; b = 0; }
                    }
                }
            }
            bitbuf |= (uint32_t)b << (24 - bitcnt);
            bitcnt += 8;
        }
    }
    int getbits(int n) {
        if (n == 0) return 0;
        if (bitcnt < n) fill();
        const int v = (int)(bitbuf >> (32 - n));
        bitbuf <<= n; bitcnt -= n;
        return v;
    }
    int getbit() { return getbits(1); }
    static int extend(int v, int n) { return v < (1 << (n - 1)) ? v - (1 << n) + 1 : v; }
    int receive_extend(int n) { return n ? extend(getbits(n), n) : 0; }
    int decode(const Huff &h) {
        if (bitcnt < 16) fill();
        const int f = h.fast[bitbuf >> 23];
        if (f >= 0) { const int s = h.size[f]; bitbuf <<= s; bitcnt -= s; return h.value[f]; }
        const int t = (int)(bitbuf >> 16);
        int l = 10;
        while (t >= h.maxcode[l]) ++l;
        if (l == 17) fail("jpeg: bad huffman code");
        const int idx = (int)(bitbuf >> (32 - l)) + h.delta[l];
        if (idx < 0 || idx >= 256) fail("jpeg: bad huffman code");
        bitbuf <<= l; bitcnt -= l;
        return h.value[idx];
    }
    void reset_scan_state() {
        bitbuf = 0; bitcnt = 0; marker = -1; no_more = false; eobrun = 0;
        for (int c = 0; c < ncomp; ++c) comp[c].dc_pred = 0;
    }

    // ---- block decoders (coefficients in natural order)
    void block_baseline(Component &c, int16_t *d) {
        const Huff &dc = hdc[c.td], &ac = hac[c.ta];
        memset(d, 0, 128);
        const int t = decode(dc);
        if (t > 15) fail("jpeg: bad DC category");
        c.dc_pred += receive_extend(t);
        d[0] = (int16_t)c.dc_pred;
        for (int k = 1; k < 64;) {
            const int rs = decode(ac), r = rs >> 4, s = rs & 15;
            if (s == 0) { if (r != 15) break; k += 16; continue; }
            k += r;
            d[kZigzag[k++]] = (int16_t)receive_extend(s);
        }
    }
    void block_dc_prog(Component &c, int16_t *d) {
        if (ah == 0) {
            const int t = decode(hdc[c.td]);
            if (t > 15) fail("jpeg: bad DC category");
            c.dc_pred += receive_extend(t);
            d[0] = (int16_t)(c.dc_pred * (1 << al));
        } else if (getbit()) d[0] = (int16_t)(d[0] + (1 << al));
    }
    void block_ac_prog(Component &c, int16_t *d) {
        const Huff &ac = hac[c.ta];
        if (ah == 0) {                           // first pass over this band
            if (eobrun) { --eobrun; return; }
            for (int k = ss; k <= se;) {
                const int rs = decode(ac), r = rs >> 4, s = rs & 15;
                if (s == 0) {
                    if (r < 15) { eobrun = (1 << r) - 1; if (r) eobrun += getbits(r); break; }
                    k += 16;
                } else {
                    k += r;
                    d[kZigzag[k++]] = (int16_t)(receive_extend(s) * (1 << al));
                }
            }
            return;
        }
        const int bit = 1 << al;                 // refinement: one more bit for known coefficients, new +-1 << al ones in between
        auto refine = [&](int16_t &v) {
            if (getbit() && (v & bit) == 0) v = (int16_t)(v > 0 ? v + bit : v - bit);
        };
        if (eobrun) {
            --eobrun;
            for (int k = ss; k <= se; ++k) { int16_t &v = d[kZigzag[k]]; if (v) refine(v); }
            return;
        }
        int k = ss;
        while (k <= se) {
            const int rs = decode(ac);
            int r = rs >> 4;
            const int s = rs & 15;
            int newval = 0;
            if (s == 0) {
                if (r < 15) {                    // end of band: the remaining known coefficients of this block still get their bit
                    eobrun = (1 << r) - 1; if (r) eobrun += getbits(r);
                    r = 64;
                }
            } else {
                if (s != 1) fail("jpeg: bad refinement code");
                newval = getbit() ? bit : -bit;
            }
            while (k <= se) {
                int16_t &v = d[kZigzag[k++]];
                if (v) refine(v);
                else {
                    if (r == 0) { v = (int16_t)newval; break; }
                    --r;
                }
            }
        }
    }

    void read_frame(bool prog) {
        progressive = prog;
        const int len = u16();
        if (u8() != 8) fail("jpeg: only 8-bit samples are supported");
        height = u16(); width = u16(); ncomp = u8();
        if (width <= 0 || height <= 0) fail("jpeg: zero-sized image");
        if (ncomp != 1 && ncomp != 3 && ncomp != 4) fail("jpeg: only 1-, 3- and 4-component images are supported");
        if (len != 8 + 3 * ncomp) fail("jpeg: bad SOF length");
        // the coefficient buffers are allocated from these two numbers before any scan data has been seen: refuse sizes no JPEG of this
        // many bytes can encode (a block costs at least a few bits; 1024 pixels per file byte is far beyond any real stream)
        if ((int64_t)width * height > (int64_t)1 << 28) fail("jpeg: image too large");
        if ((int64_t)width * height > (int64_t)1 << 24 && (int64_t)width * height > (int64_t)(end - p) * 1024) fail("jpeg: image size implausible for the file size");
        for (int i = 0; i < ncomp; ++i) {
            Component &c = comp[i];
            c.id = u8(); const int hv = u8(); c.h = hv >> 4; c.v = hv & 15; c.tq = u8();
            if (c.h < 1 || c.h > 2 || c.v < 1 || c.v > 2 || c.tq > 3) fail("jpeg: unsupported sampling factors");
            hmax = c.h > hmax ? c.h : hmax; vmax = c.v > vmax ? c.v : vmax;
        }
        const int mcux = (width + 8 * hmax - 1) / (8 * hmax), mcuy = (height + 8 * vmax - 1) / (8 * vmax);
        for (int i = 0; i < ncomp; ++i) {
            Component &c = comp[i];
            c.bw = mcux * c.h; c.bh = mcuy * c.v; c.pw = c.bw * 8; c.ph = c.bh * 8;
            const int cw = (width * c.h + hmax - 1) / hmax, ch = (height * c.v + vmax - 1) / vmax;
            c.nbx = (cw + 7) / 8; c.nby = (ch + 7) / 8;
            c.coef.assign((size_t)c.bw * c.bh * 64, 0);
        }
    }
    void read_dqt() {
        int len = u16() - 2;
        while (len > 0) {
            const int pq = u8(), prec = pq >> 4, t = pq & 15;
            if (t > 3 || prec > 1) fail("jpeg: bad DQT");
            for (int i = 0; i < 64; ++i) qt[t][kZigzag[i]] = (uint16_t)(prec ? u16() : u8());
            qt_present[t] = true;
            len -= 65 + 64 * prec;
        }
        if (len != 0) fail("jpeg: bad DQT length");
    }
    void read_dht() {
        int len = u16() - 2;
        while (len > 0) {
            const int tc = u8(), cls = tc >> 4, t = tc & 15;
            if (cls > 1 || t > 3) fail("jpeg: bad DHT");
            uint8_t counts[16]; int n = 0;
            for (int i = 0; i < 16; ++i) { counts[i] = (uint8_t)u8(); n += counts[i]; }
            if (n > 256) fail("jpeg: bad DHT");
            uint8_t vals[256];
            for (int i = 0; i < n; ++i) vals[i] = (uint8_t)u8();
            (cls ? hac[t] : hdc[t]).build(counts, vals, n);
            len -= 17 + n;
        }
        if (len != 0) fail("jpeg: bad DHT length");
    }
    void read_scan() {
        const int len = u16();
        const int ns = u8();
        if (ns < 1 || ns > ncomp || len != 6 + 2 * ns) fail("jpeg: bad SOS");
        int order[4];
        for (int i = 0; i < ns; ++i) {
            const int id = u8(), tt = u8();
            int k = -1;
            for (int c = 0; c < ncomp; ++c) if (comp[c].id == id) k = c;
            if (k < 0) fail("jpeg: SOS names an unknown component");
            comp[k].td = tt >> 4; comp[k].ta = tt & 15;
            if (comp[k].td > 3 || comp[k].ta > 3) fail("jpeg: bad table selector");
            order[i] = k;
        }
        ss = u8(); se = u8(); const int a = u8(); ah = a >> 4; al = a & 15;
        if (progressive) { if (ss > 63 || se > 63 || ss > se || ah > 13 || al > 13 || (ss == 0 && se != 0) || (ss > 0 && ns != 1)) fail("jpeg: bad progressive scan parameters"); }
        else { if (ss != 0 || se != 63 || ah != 0 || al != 0) fail("jpeg: bad baseline scan parameters"); }
        for (int i = 0; i < ns; ++i) {
            const Component &c = comp[order[i]];
            if ((!progressive || ss == 0) && ah == 0 && !hdc[c.td].present) fail("jpeg: missing DC huffman table");
            if ((!progressive || ss > 0) && !hac[c.ta].present) fail("jpeg: missing AC huffman table");
        }
        reset_scan_state();
        int todo = restart_interval ? restart_interval : 0x7fffffff;
        auto after_mcu = [&]() {
            if (--todo > 0) return;
            // restart interval: byte-align, expect RSTn, reset predictors
            if (bitcnt < 24) fill();
            if (marker >= 0xD0 && marker <= 0xD7) { reset_scan_state(); todo = restart_interval; }
            else todo = 0x7fffffff;              // no (more) restart markers: keep decoding, the end-of-scan marker stops us
        };
        auto one_block = [&](Component &c, int bx, int by) {
            int16_t *d = &c.coef[((size_t)by * c.bw + bx) * 64];
            if (!progressive) block_baseline(c, d);
            else if (ss == 0) block_dc_prog(c, d);
            else block_ac_prog(c, d);
        };
        if (ns == 1) {                           // non-interleaved: the component's own block grid, clipped to the image
            Component &c = comp[order[0]];
            for (int by = 0; by < c.nby; ++by)
                for (int bx = 0; bx < c.nbx; ++bx) { one_block(c, bx, by); after_mcu(); }
        } else {
            const int mcux = comp[0].bw / comp[0].h, mcuy = comp[0].bh / comp[0].v;
            for (int my = 0; my < mcuy; ++my)
                for (int mx = 0; mx < mcux; ++mx) {
                    for (int i = 0; i < ns; ++i) {
                        Component &c = comp[order[i]];
                        for (int y = 0; y < c.v; ++y)
                            for (int x = 0; x < c.h; ++x) one_block(c, mx * c.h + x, my * c.v + y);
                    }
                    after_mcu();
                }
        }
        // position after the scan: the bit reader stopped at a marker (or ran out); rewind to it
        if (marker >= 0) { p -= 2; }
    }

    // stb_image's integer IDCT (jidctint "islow" with 12-bit constants), output clamped to u8 with the +128 level shift folded in
    static void idct(const int16_t *in, const uint16_t *q, uint8_t *out, int stride) {
        // 64-bit intermediates: identical results on every valid stream; corrupt coefficients (int16 x 16-bit DQT entry x 4096)
        // cannot overflow (r02 advisor: signed overflow is undefined behaviour)
        int64_t val[64];
        auto f2f = [](double x) { return (int)(x * 4096 + 0.5); };
        static const int c0541 = f2f(0.5411961), c1847 = f2f(-1.847759065), c0765 = f2f(0.765366865), c1175 = f2f(1.175875602), c0298 = f2f(0.298631336),
                         c2053 = f2f(2.053119869), c3072 = f2f(3.072711026), c1501 = f2f(1.501321110), c0899 = f2f(-0.899976223), c2562 = f2f(-2.562915447),
                         c1961 = f2f(-1.961570560), c0390 = f2f(-0.390180644);
#define VITX_IDCT_1D(s0, s1, s2, s3, s4, s5, s6, s7)                                                                          \
        int64_t t0, t1, t2, t3, p1, p2, p3, p4, p5, x0, x1, x2, x3;                                                               \
        p2 = s2; p3 = s6; p1 = (p2 + p3) * c0541; t2 = p1 + p3 * c1847; t3 = p1 + p2 * c0765;                                     \
        p2 = s0; p3 = s4; t0 = (p2 + p3) * 4096; t1 = (p2 - p3) * 4096;                                                           \
        x0 = t0 + t3; x3 = t0 - t3; x1 = t1 + t2; x2 = t1 - t2;                                                                   \
        t0 = s7; t1 = s5; t2 = s3; t3 = s1; p3 = t0 + t2; p4 = t1 + t3; p1 = t0 + t3; p2 = t1 + t2; p5 = (p3 + p4) * c1175;       \
        t0 = t0 * c0298; t1 = t1 * c2053; t2 = t2 * c3072; t3 = t3 * c1501;                                                       \
        p1 = p5 + p1 * c0899; p2 = p5 + p2 * c2562; p3 = p3 * c1961; p4 = p4 * c0390;                                             \
        t3 += p1 + p4; t2 += p2 + p3; t1 += p2 + p4; t0 += p1 + p3;
        for (int i = 0; i < 8; ++i) {
            const int16_t *d = in + i; const uint16_t *dq = q + i; int64_t *v = val + i;
            if (d[8] == 0 && d[16] == 0 && d[24] == 0 && d[32] == 0 && d[40] == 0 && d[48] == 0 && d[56] == 0) {
                const int64_t dc = (int64_t)d[0] * dq[0] * 4;
                v[0] = v[8] = v[16] = v[24] = v[32] = v[40] = v[48] = v[56] = dc;
            } else {
                VITX_IDCT_1D((int64_t)d[0] * dq[0], (int64_t)d[8] * dq[8], (int64_t)d[16] * dq[16], (int64_t)d[24] * dq[24], (int64_t)d[32] * dq[32], (int64_t)d[40] * dq[40], (int64_t)d[48] * dq[48], (int64_t)d[56] * dq[56])
                x0 += 512; x1 += 512; x2 += 512; x3 += 512;
                v[0] = (x0 + t3) >> 10; v[56] = (x0 - t3) >> 10; v[8] = (x1 + t2) >> 10; v[48] = (x1 - t2) >> 10;
                v[16] = (x2 + t1) >> 10; v[40] = (x2 - t1) >> 10; v[24] = (x3 + t0) >> 10; v[32] = (x3 - t0) >> 10;
            }
        }
        auto clamp = [](int64_t x) { return (uint8_t)(x < 0 ? 0 : (x > 255 ? 255 : x)); };
        for (int i = 0; i < 8; ++i) {
            const int64_t *v = val + i * 8; uint8_t *o = out + i * stride;
            VITX_IDCT_1D(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7])
            x0 += 65536 + (128 << 17); x1 += 65536 + (128 << 17); x2 += 65536 + (128 << 17); x3 += 65536 + (128 << 17);      // fits: |x| < 2^40
            o[0] = clamp((x0 + t3) >> 17); o[7] = clamp((x0 - t3) >> 17); o[1] = clamp((x1 + t2) >> 17); o[6] = clamp((x1 - t2) >> 17);
            o[2] = clamp((x2 + t1) >> 17); o[5] = clamp((x2 - t1) >> 17); o[3] = clamp((x3 + t0) >> 17); o[4] = clamp((x3 - t0) >> 17);
        }
#undef VITX_IDCT_1D
    }

    // chroma row resamplers (near = the closer source row, far = the other one)
    static void up_1(uint8_t *out, const uint8_t *near, const uint8_t *, int w) { memcpy(out, near, w); }
    static void up_v2(uint8_t *out, const uint8_t *near, const uint8_t *far, int w) { for (int i = 0; i < w; ++i) out[i] = (uint8_t)((3 * near[i] + far[i] + 2) >> 2); }
    static void up_h2(uint8_t *out, const uint8_t *in, const uint8_t *, int w) {
        if (w == 1) { out[0] = out[1] = in[0]; return; }
        out[0] = in[0]; out[1] = (uint8_t)((in[0] * 3 + in[1] + 2) >> 2);
        int i;
        for (i = 1; i < w - 1; ++i) { const int n = 3 * in[i] + 2; out[i * 2] = (uint8_t)((n + in[i - 1]) >> 2); out[i * 2 + 1] = (uint8_t)((n + in[i + 1]) >> 2); }
        out[i * 2] = (uint8_t)((in[w - 2] * 3 + in[w - 1] + 2) >> 2); out[i * 2 + 1] = in[w - 1];
    }
    static void up_hv2(uint8_t *out, const uint8_t *near, const uint8_t *far, int w) {
        if (w == 1) { out[0] = out[1] = (uint8_t)((3 * near[0] + far[0] + 2) >> 2); return; }
        int t1 = 3 * near[0] + far[0];
        out[0] = (uint8_t)((t1 + 2) >> 2);
        for (int i = 1; i < w; ++i) {
            const int t0 = t1; t1 = 3 * near[i] + far[i];
            out[i * 2 - 1] = (uint8_t)((3 * t0 + t1 + 8) >> 4); out[i * 2] = (uint8_t)((3 * t1 + t0 + 8) >> 4);
        }
        out[w * 2 - 1] = (uint8_t)((t1 + 2) >> 2);
    }

    std::vector<uint8_t> decode_all() {
        if (u8() != 0xFF || u8() != 0xD8) fail("jpeg: no SOI");
        bool have_frame = false, done = false;
        while (!done) {
            int m = u8();
            if (m != 0xFF) continue;             // tolerate stray bytes between segments
            while (m == 0xFF) m = u8();
            switch (m) {
            case 0xC0: case 0xC1: if (have_frame) fail("jpeg: two frames"); read_frame(false); have_frame = true; break;
            case 0xC2: if (have_frame) fail("jpeg: two frames"); read_frame(true); have_frame = true; break;
            case 0xC3: case 0xC5: case 0xC6: case 0xC7: case 0xC9: case 0xCA: case 0xCB: case 0xCD: case 0xCE: case 0xCF:
                fail("jpeg: lossless, hierarchical and arithmetic-coded files are not supported");
            case 0xC4: read_dht(); break;
            case 0xDB: read_dqt(); break;
            case 0xDD: if (u16() != 4) fail("jpeg: bad DRI"); restart_interval = u16(); break;
            case 0xDA: if (!have_frame) fail("jpeg: scan before frame"); read_scan(); break;
            case 0xD9: done = true; break;
            case 0x00: break;
            default:
                if (m >= 0xD0 && m <= 0xD7) break;               // stray restart marker
                {
                    const int len = u16(); if (len < 2 || p + len - 2 > end) fail("jpeg: truncated segment");
                    if (m == 0xE0 && len >= 7 && !memcmp(p, "JFIF", 5)) jfif = true;
                    if (m == 0xEE && len >= 14 && !memcmp(p, "Adobe", 5)) app14_transform = p[11];       // version(2) flags0(2) flags1(2) transform(1)
                    p += len - 2;
                }
            }
            if (p >= end) done = true;
        }
        if (!have_frame) fail("jpeg: no frame");
        for (int i = 0; i < ncomp; ++i) {
            Component &c = comp[i];
            if (!qt_present[c.tq]) fail("jpeg: missing quantisation table");
            c.pix.assign((size_t)c.pw * c.ph, 0);
            for (int by = 0; by < c.bh; ++by)
                for (int bx = 0; bx < c.bw; ++bx) idct(&c.coef[((size_t)by * c.bw + bx) * 64], qt[c.tq], &c.pix[(size_t)by * 8 * c.pw + bx * 8], c.pw);
            std::vector<int16_t>().swap(c.coef);
        }
        std::vector<uint8_t> rgb((size_t)width * height * 3);
        if (ncomp == 1) {
            for (int y = 0; y < height; ++y)
                for (int x = 0; x < width; ++x) { const uint8_t g = comp[0].pix[(size_t)y * comp[0].pw + x]; uint8_t *o = &rgb[((size_t)y * width + x) * 3]; o[0] = o[1] = o[2] = g; }
            return rgb;
        }
        struct Res { void (*fn)(uint8_t *, const uint8_t *, const uint8_t *, int); int hs, vs, ystep, ypos, w_lores; const uint8_t *line0, *line1; std::vector<uint8_t> buf; } r[4];
        for (int k = 0; k < ncomp; ++k) {
            Component &c = comp[k];
            r[k].hs = hmax / c.h; r[k].vs = vmax / c.v; r[k].ystep = r[k].vs >> 1; r[k].ypos = 0;
            r[k].w_lores = (width + r[k].hs - 1) / r[k].hs;
            r[k].line0 = r[k].line1 = c.pix.data();
            r[k].buf.resize((size_t)width + 8);
            r[k].fn = (r[k].hs == 1 && r[k].vs == 1) ? up_1 : (r[k].hs == 1 ? up_v2 : (r[k].vs == 1 ? up_h2 : up_hv2));
        }
        auto f2f = [](float x) { return ((int)(x * 4096.0f + 0.5f)) << 8; };
        const int cr_r = f2f(1.40200f), cr_g = -f2f(0.71414f), cb_g = -f2f(0.34414f), cb_b = f2f(1.77200f);
        auto clamp = [](int x) { return (uint8_t)((unsigned)x > 255 ? (x < 0 ? 0 : 255) : x); };
        auto blinn = [](int x, int y) { const unsigned t = (unsigned)(x * y + 128); return (uint8_t)((t + (t >> 8)) >> 8); };     // x * y / 255, rounded
        // what stbi_load(..., 3) does with the components: three components are RGB when they are named 'R','G','B' or an Adobe marker
        // says "no transform" (and no JFIF marker contradicts it), otherwise YCbCr; four are CMYK (transform 0), YCCK (2) or YCbCr + an
        // ignored fourth; Adobe writes CMYK inverted, hence the products with k
        const bool named_rgb = ncomp == 3 && comp[0].id == 'R' && comp[1].id == 'G' && comp[2].id == 'B';
        const bool is_rgb = ncomp == 3 && (named_rgb || (app14_transform == 0 && !jfif));
        const uint8_t *row[4];
        for (int j = 0; j < height; ++j) {
            for (int k = 0; k < ncomp; ++k) {
                Res &q = r[k];
                const bool y_bot = q.ystep >= (q.vs >> 1);
                if (q.fn == up_1) row[k] = y_bot ? q.line1 : q.line0;
                else { q.fn(q.buf.data(), y_bot ? q.line1 : q.line0, y_bot ? q.line0 : q.line1, q.w_lores); row[k] = q.buf.data(); }
                if (++q.ystep >= q.vs) {
                    q.ystep = 0; q.line0 = q.line1;
                    const int comp_rows = (height * comp[k].v + vmax - 1) / vmax;
                    if (++q.ypos < comp_rows) q.line1 += comp[k].pw;
                }
            }
            uint8_t *o = &rgb[(size_t)j * width * 3];
            for (int i = 0; i < width; ++i) {
                if (is_rgb) { o[i * 3] = row[0][i]; o[i * 3 + 1] = row[1][i]; o[i * 3 + 2] = row[2][i]; continue; }
                if (ncomp == 4 && app14_transform == 0) {       // CMYK
                    const int k = row[3][i];
                    o[i * 3] = blinn(row[0][i], k); o[i * 3 + 1] = blinn(row[1][i], k); o[i * 3 + 2] = blinn(row[2][i], k);
                    continue;
                }
                const int yf = (row[0][i] << 20) + (1 << 19), cb = row[1][i] - 128, cr = row[2][i] - 128;
                int rr = yf + cr * cr_r, gg = yf + cr * cr_g + ((cb * cb_g) & 0xffff0000), bb = yf + cb * cb_b;
                o[i * 3] = clamp(rr >> 20); o[i * 3 + 1] = clamp(gg >> 20); o[i * 3 + 2] = clamp(bb >> 20);
                if (ncomp == 4 && app14_transform == 2) {       // YCCK
                    const int k = row[3][i];
                    o[i * 3] = blinn(255 - o[i * 3], k); o[i * 3 + 1] = blinn(255 - o[i * 3 + 1], k); o[i * 3 + 2] = blinn(255 - o[i * 3 + 2], k);
                }
            }
        }
        return rgb;
    }
};

// ------------------------------------------------------------------------------------------------------------------
// zlib inflate (RFC 1950 / 1951) and PNG
// ------------------------------------------------------------------------------------------------------------------
struct Inflate {
    const uint8_t *p, *end; uint32_t bits = 0; int nbits = 0;
    std::vector<uint8_t> out;
    size_t limit = (size_t)-1;      // the caller knows how many bytes a valid stream expands to: anything beyond is a decompression bomb
    void grown() { if (out.size() > limit) fail("png: the zlib stream expands beyond the image size"); }
    struct Tab { uint16_t count[16]; uint16_t sym[288]; };
    int bit() { if (!nbits) { if (p >= end) fail("png: truncated zlib stream"); bits = *p++; nbits = 8; } const int b = bits & 1; bits >>= 1; --nbits; return b; }
    int get(int n) { int v = 0; for (int i = 0; i < n; ++i) v |= bit() << i; return v; }
    static void build(Tab &t, const uint8_t *len, int n) {
        memset(t.count, 0, sizeof t.count);
        for (int i = 0; i < n; ++i) t.count[len[i]]++;
        t.count[0] = 0;
        uint16_t offs[16]; offs[1] = 0;
        for (int i = 1; i < 15; ++i) offs[i + 1] = (uint16_t)(offs[i] + t.count[i]);
        for (int i = 0; i < n; ++i) if (len[i]) t.sym[offs[len[i]]++] = (uint16_t)i;
    }
    int sym(const Tab &t) {
        int code = 0, first = 0, index = 0;
        for (int l = 1; l <= 15; ++l) {
            code |= bit();
            const int c = t.count[l];
            if (code - c < first) return t.sym[index + (code - first)];
            index += c; first += c; first <<= 1; code <<= 1;
        }
        fail("png: bad huffman code");
    }
    void codes(const Tab &lit, const Tab &dist) {
        static const uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
        static const uint8_t lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
        static const uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
        static const uint8_t dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
        for (;;) {
            int s = sym(lit);
            if (s < 256) { out.push_back((uint8_t)s); grown(); continue; }
            if (s == 256) return;
            s -= 257;
            if (s >= 29) fail("png: bad length code");
            const int len = lbase[s] + get(lext[s]);
            const int ds = sym(dist);
            if (ds >= 30) fail("png: bad distance code");
            const size_t d = dbase[ds] + (size_t)get(dext[ds]);
            if (d > out.size()) fail("png: distance beyond the window");
            const size_t from = out.size() - d;
            for (int i = 0; i < len; ++i) out.push_back(out[from + i]);
            grown();
        }
    }
    void run() {
        if (end - p < 2) fail("png: truncated zlib header");
        const int cmf = p[0], flg = p[1]; p += 2;
        if ((cmf & 15) != 8 || ((cmf << 8) | flg) % 31 || (flg & 32)) fail("png: bad zlib header");
        int last;
        do {
            last = bit();
            const int type = get(2);
            if (type == 0) {
                nbits = 0;
                if (end - p < 4) fail("png: truncated stored block");
                const int len = p[0] | (p[1] << 8), nlen = p[2] | (p[3] << 8); p += 4;
                if ((len ^ 0xffff) != nlen || end - p < len) fail("png: bad stored block");
                out.insert(out.end(), p, p + len); p += len; grown();
            } else if (type == 1) {
                uint8_t l[288]; for (int i = 0; i < 144; ++i) l[i] = 8; for (int i = 144; i < 256; ++i) l[i] = 9; for (int i = 256; i < 280; ++i) l[i] = 7; for (int i = 280; i < 288; ++i) l[i] = 8;
                uint8_t d[30]; for (int i = 0; i < 30; ++i) d[i] = 5;
                Tab lt, dt; build(lt, l, 288); build(dt, d, 30); codes(lt, dt);
            } else if (type == 2) {
                const int nlen = get(5) + 257, ndist = get(5) + 1, ncode = get(4) + 4;
                if (nlen > 286 || ndist > 30) fail("png: bad dynamic block");
                static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                uint8_t cl[19] = {0};
                for (int i = 0; i < ncode; ++i) cl[order[i]] = (uint8_t)get(3);
                Tab ct; build(ct, cl, 19);
                uint8_t lens[320]; int n = 0;
                while (n < nlen + ndist) {
                    const int s = sym(ct);
                    if (s < 16) lens[n++] = (uint8_t)s;
                    else {
                        int prev = 0, rep;
                        if (s == 16) { if (!n) fail("png: repeat without a previous length"); prev = lens[n - 1]; rep = 3 + get(2); }
                        else if (s == 17) rep = 3 + get(3);
                        else rep = 11 + get(7);
                        if (n + rep > nlen + ndist) fail("png: too many code lengths");
                        while (rep--) lens[n++] = (uint8_t)prev;
                    }
                }
                if (lens[256] == 0) fail("png: no end-of-block code");
                Tab lt, dt; build(lt, lens, nlen); build(dt, lens + nlen, ndist); codes(lt, dt);
            } else fail("png: bad block type");
        } while (!last);
    }
};

std::vector<uint8_t> decode_png(const uint8_t *data, size_t n, int &w, int &h) {
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (n < 8 || memcmp(data, sig, 8)) fail("png: bad signature");
    const uint8_t *p = data + 8, *end = data + n;
    int depth = 0, ctype = -1, interlace = 0;
    std::vector<uint8_t> idat, plte;
    bool seen_end = false;
    while (!seen_end) {
        if (end - p < 12) fail("png: truncated chunk");
        const uint32_t len = ((uint32_t)p[0] << 24) | (p[1] << 16) | (p[2] << 8) | p[3];
        const char *tag = (const char *)p + 4;
        if ((size_t)(end - p) < 12 + (size_t)len) fail("png: truncated chunk");
        const uint8_t *d = p + 8;
        if (!memcmp(tag, "IHDR", 4)) {
            if (len != 13) fail("png: bad IHDR");
            w = (d[0] << 24) | (d[1] << 16) | (d[2] << 8) | d[3]; h = (d[4] << 24) | (d[5] << 16) | (d[6] << 8) | d[7];
            depth = d[8]; ctype = d[9]; interlace = d[12];
            if (w <= 0 || h <= 0 || (int64_t)w * h > (int64_t)1 << 28) fail("png: bad image size");
            if (d[10] != 0 || d[11] != 0) fail("png: bad compression / filter method");
        } else if (!memcmp(tag, "PLTE", 4)) plte.assign(d, d + len);
        else if (!memcmp(tag, "IDAT", 4)) idat.insert(idat.end(), d, d + len);
        else if (!memcmp(tag, "IEND", 4)) seen_end = true;
        p += 12 + len;
    }
    if (ctype < 0) fail("png: no IHDR");
    if (interlace > 1) fail("png: unknown interlace method");
    int channels = 0;
    switch (ctype) { case 0: channels = 1; break; case 2: channels = 3; break; case 3: channels = 1; break; case 4: channels = 2; break; case 6: channels = 4; break; default: fail("png: bad colour type"); }
    if (!(depth == 8 || depth == 16 || (depth < 8 && (ctype == 0 || ctype == 3) && (depth == 1 || depth == 2 || depth == 4)))) fail("png: unsupported bit depth");
    if (ctype == 3 && (depth == 16 || plte.size() < 3)) fail("png: bad palette image");
    const size_t bpp = (size_t)(channels * depth + 7) / 8;
    auto row_bytes = [&](int pw) { return ((size_t)pw * channels * depth + 7) / 8; };
    // Adam7 (RFC 2083 section 2.6): seven reduced images, each filtered on its own; pass p holds pixels (xo + i * xs, yo + j * ys)
    static const int xo[7] = {0, 4, 0, 2, 0, 1, 0}, yo[7] = {0, 0, 4, 0, 2, 0, 1}, xs[7] = {8, 8, 4, 4, 2, 2, 1}, ys[7] = {8, 8, 8, 4, 4, 2, 2};
    size_t need = 0;
    if (!interlace) need = (row_bytes(w) + 1) * (size_t)h;
    else for (int ps = 0; ps < 7; ++ps) { const int pw = (w - xo[ps] + xs[ps] - 1) / xs[ps], ph = (h - yo[ps] + ys[ps] - 1) / ys[ps]; if (pw > 0 && ph > 0) need += (row_bytes(pw) + 1) * (size_t)ph; }
    Inflate z; z.p = idat.data(); z.end = idat.data() + idat.size();
    z.limit = need;                                                   // a valid stream expands to exactly `need` bytes
    z.out.reserve(need < ((size_t)64 << 20) ? need : ((size_t)64 << 20));
    z.run();
    if (z.out.size() < need) fail("png: not enough image data");
    std::vector<uint8_t> rgb((size_t)w * h * 3);
    // un-filter a pw x ph (reduced) image starting at src and hand every pixel to put(x, y, r, g, b)
    auto unfilter = [&](const uint8_t *src, int pw, int ph, auto &&put) {
        const size_t stride = row_bytes(pw);
        std::vector<uint8_t> cur(stride), prev(stride, 0);
        for (int y = 0; y < ph; ++y) {
            const int ft = src[0]; ++src;
            for (size_t i = 0; i < stride; ++i) {
                const int a = i >= bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0;
                int v;
                switch (ft) {
                case 0: v = src[i]; break;
                case 1: v = src[i] + a; break;
                case 2: v = src[i] + b; break;
                case 3: v = src[i] + ((a + b) >> 1); break;
                case 4: { const int pa = abs(b - c), pb = abs(a - c), pc = abs(a + b - 2 * c); v = src[i] + ((pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c)); break; }
                default: fail("png: bad filter type");
                }
                cur[i] = (uint8_t)v;
            }
            src += stride;
            for (int x = 0; x < pw; ++x) {
                auto sample = [&](int ch) -> int {      // channel ch of pixel x scaled to 8 bits
                    if (depth == 8) return cur[(size_t)x * channels + ch];
                    if (depth == 16) return cur[((size_t)x * channels + ch) * 2];
                    const int per = 8 / depth, v = (cur[x / per] >> ((per - 1 - x % per) * depth)) & ((1 << depth) - 1);
                    return ctype == 3 ? v : v * 255 / ((1 << depth) - 1);
                };
                if (ctype == 3) { const size_t i = (size_t)sample(0) * 3; if (i + 3 > plte.size()) fail("png: palette index out of range"); put(x, y, plte[i], plte[i + 1], plte[i + 2]); }
                else if (ctype == 0 || ctype == 4) { const uint8_t g = (uint8_t)sample(0); put(x, y, g, g, g); }
                else put(x, y, (uint8_t)sample(0), (uint8_t)sample(1), (uint8_t)sample(2));
            }
            prev.swap(cur);
        }
        return src;
    };
    if (!interlace) {
        unfilter(z.out.data(), w, h, [&](int x, int y, uint8_t r, uint8_t g, uint8_t b) { uint8_t *o = &rgb[((size_t)y * w + x) * 3]; o[0] = r; o[1] = g; o[2] = b; });
    } else {
        const uint8_t *src = z.out.data();
        for (int ps = 0; ps < 7; ++ps) {
            const int pw = (w - xo[ps] + xs[ps] - 1) / xs[ps], ph = (h - yo[ps] + ys[ps] - 1) / ys[ps];
            if (pw <= 0 || ph <= 0) continue;
            src = unfilter(src, pw, ph, [&](int x, int y, uint8_t r, uint8_t g, uint8_t b) {
                uint8_t *o = &rgb[((size_t)(yo[ps] + y * ys[ps]) * w + (xo[ps] + x * xs[ps])) * 3]; o[0] = r; o[1] = g; o[2] = b; });
        }
    }
    return rgb;
}

std::vector<uint8_t> decode_ppm(const uint8_t *data, size_t n, int &w, int &h) {
    size_t pos = 2; int vals[3], got = 0;
    while (got < 3) {
        while (pos < n && (data[pos] == ' ' || data[pos] == '\n' || data[pos] == '\r' || data[pos] == '\t')) ++pos;
        if (pos < n && data[pos] == '#') { while (pos < n && data[pos] != '\n') ++pos; continue; }
        if (pos >= n || data[pos] < '0' || data[pos] > '9') fail("ppm: bad header");
        int v = 0; while (pos < n && data[pos] >= '0' && data[pos] <= '9') { v = v * 10 + (data[pos] - '0'); if (v > (1 << 28)) fail("ppm: bad header"); ++pos; }
        vals[got++] = v;
    }
    // exactly one whitespace byte separates maxval from the raster; a file that ends with the header has none (r02 advisor finding:
    // the unconditional ++pos moved past the end and the unsigned `n - pos` below wrapped around)
    if (pos >= n || !(data[pos] == ' ' || data[pos] == '\n' || data[pos] == '\r' || data[pos] == '\t')) fail("ppm: truncated (no raster after the header)");
    ++pos;
    w = vals[0]; h = vals[1];
    if (w <= 0 || h <= 0 || vals[2] != 255 || (int64_t)w * h > (int64_t)1 << 28) fail("ppm: only 8-bit P6 files are supported");
    if (n - pos < (size_t)w * h * 3) fail("ppm: truncated");
    return std::vector<uint8_t>(data + pos, data + pos + (size_t)w * h * 3);
}

}  // namespace

extern "C" {

int vitx_image_decode(const uint8_t *bytes, size_t n, uint8_t **out_rgb, int *nx, int *ny) {
    if (!bytes || !out_rgb || !nx || !ny) { vitx::set_error("vitx_image_decode: NULL argument"); return VITX_ERR_ARG; }
    *out_rgb = nullptr; *nx = *ny = 0;
    try {
        std::vector<uint8_t> rgb; int w = 0, h = 0;
        if (n >= 3 && bytes[0] == 0xFF && bytes[1] == 0xD8) { Jpeg j; j.p = bytes; j.end = bytes + n; rgb = j.decode_all(); w = j.width; h = j.height; }
        else if (n >= 8 && bytes[0] == 0x89 && bytes[1] == 'P') rgb = decode_png(bytes, n, w, h);
        else if (n >= 2 && bytes[0] == 'P' && bytes[1] == '6') rgb = decode_ppm(bytes, n, w, h);
        else { vitx::set_error("vitx_image_decode: not a JPEG, PNG or binary PPM file"); return VITX_ERR_FORMAT; }
        uint8_t *o = (uint8_t *)malloc(rgb.size() ? rgb.size() : 1);
        if (!o) return VITX_ERR_NOMEM;
        memcpy(o, rgb.data(), rgb.size());
        *out_rgb = o; *nx = w; *ny = h;
        return VITX_OK;
    } catch (const DecodeError &e) { vitx::set_error("vitx_image_decode: %s", e.msg.c_str()); return VITX_ERR_FORMAT; }
    catch (const std::bad_alloc &) { vitx::set_error("vitx_image_decode: out of memory"); return VITX_ERR_NOMEM; }
}

int vitx_image_load(const char *path, uint8_t **out_rgb, int *nx, int *ny) {
    if (!path) { vitx::set_error("vitx_image_load: NULL path"); return VITX_ERR_ARG; }
    FILE *f = fopen(path, "rb");
    if (!f) { vitx::set_error("vitx_image_load: failed to open '%s'", path); return VITX_ERR_IO; }
    std::vector<uint8_t> buf;
    uint8_t tmp[65536]; size_t got;
    while ((got = fread(tmp, 1, sizeof tmp, f)) > 0) buf.insert(buf.end(), tmp, tmp + got);
    fclose(f);
    return vitx_image_decode(buf.data(), buf.size(), out_rgb, nx, ny);
}

void vitx_image_free(uint8_t *rgb) { free(rgb); }

}  // extern "C"
