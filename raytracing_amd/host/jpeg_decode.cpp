// jpeg_decode.cpp — JPEG (ITU-T T.81) decoder of the host surface: baseline and progressive Huffman streams, 8 bits per
// sample, grey or YCbCr, sampling factors 1 or 2 -> 8-bit RGB.
//
// The reference reads textures through lodepng only (imageio.cpp:7-45), but the model it ships keeps its textures as JPEG
// files (DamagedHelmet: Default_albedo.jpg baseline 4:2:0, Default_metalRoughness.jpg progressive 4:4:4; earthmap.jpg
// baseline 4:4:4), so the loader needs a JPEG path of its own.  Nothing here comes from the reference.  The arithmetic
// follows the Independent JPEG Group's decoder bit for bit, so that texels equal what libjpeg / libjpeg-turbo based tools
// (the ones used to make PNG twins of these files) produce:
//   * inverse DCT: the "islow" 13-bit fixed-point Loeffler-Ligtenberg-Moschytz transform, two passes, PASS1_BITS = 2;
//   * chroma upsampling: the "fancy" triangle filters (3/4, 1/4 taps; h2v1 and h2v2), edge samples replicated;
//   * YCbCr -> RGB: 16-bit fixed-point tables (1.402, 1.772, 0.71414, 0.34414), rounding as in jdcolor.c.
#include <cstdint>
#include <cstring>
#include <exception>
#include <vector>

#include "rt_host.h"

namespace {

const uint8_t kZigzag[64 + 16] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13,
                                  6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31,
                                  39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
                                  63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};   // guard entries for corrupt run lengths

struct Huff {
    bool present = false;
    uint8_t bits[17] = {0}, vals[256] = {0};
    int mincode[17], maxcode[18], valptr[17];
    void build() {
        int code = 0, k = 0;
        for (int l = 1; l <= 16; ++l) {
            valptr[l] = k; mincode[l] = code;
            code += bits[l]; k += bits[l];
            maxcode[l] = bits[l] ? code - 1 : -1;
            code <<= 1;
        }
        maxcode[17] = 0x7FFFFFFF;
        present = true;
    }
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0;
    int w = 0, hgt = 0;              // real samples: ceil(W * h / hmax), ceil(H * v / vmax)
    int bw = 0, bh = 0;              // blocks stored (padded to whole MCUs)
    int pred = 0;
    std::vector<int16_t> coef;       // bw * bh blocks of 64, natural order
    std::vector<uint8_t> plane;      // (bw * 8) x (bh * 8) samples after the inverse DCT
};

struct Decoder {
    const uint8_t *d; size_t n, pos = 0;
    int W = 0, H = 0, ncomp = 0, hmax = 1, vmax = 1, mcux = 0, mcuy = 0;
    bool progressive = false, ok = true;
    Component comp[3];
    uint16_t qt[4][64];
    bool have_qt[4] = {false, false, false, false};
    Huff dc[4], ac[4];
    int restart_interval = 0;
    // entropy-coded segment reader
    uint32_t bitbuf = 0; int bitcnt = 0; bool hit_marker = false;
    int eobrun = 0;

    Decoder(const uint8_t *data, size_t size) : d(data), n(size) {}

    int u8() { return pos < n ? d[pos++] : (ok = false, 0); }
    int u16() { int a = u8(); return (a << 8) | u8(); }

    void fill() {
        while (bitcnt <= 24) {
            int b = 0;
            if (!hit_marker && pos < n) {
                b = d[pos];
                if (b == 0xFF) {
                    int b2 = pos + 1 < n ? d[pos + 1] : 0xD9;
                    if (b2 == 0) pos += 2;                       // stuffed zero
                    else { hit_marker = true; b = 0; }            // a marker ends the segment: feed zeros
                } else pos++;
            }
            bitbuf |= (uint32_t)b << (24 - bitcnt);
            bitcnt += 8;
        }
    }
    int getbits(int k) {
        if (k == 0) return 0;
        if (bitcnt < k) fill();
        int v = (int)(bitbuf >> (32 - k));
        bitbuf <<= k; bitcnt -= k;
        return v;
    }
    int getbit() { return getbits(1); }
    int decode(const Huff &h) {
        if (!h.present) { ok = false; return 0; }
        int code = 0;
        for (int l = 1; l <= 16; ++l) {
            code = (code << 1) | getbit();
            if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) return h.vals[h.valptr[l] + code - h.mincode[l]];
        }
        ok = false;
        return 0;
    }
    static int extend(int v, int s) { return s == 0 ? 0 : (v < (1 << (s - 1)) ? v - (1 << s) + 1 : v); }

    void reset_entropy() { bitbuf = 0; bitcnt = 0; hit_marker = false; eobrun = 0; for (Component &c : comp) c.pred = 0; }
    void restart() {       // byte align, skip the RSTn marker
        bitbuf = 0; bitcnt = 0; hit_marker = false;
        while (pos + 1 < n && !(d[pos] == 0xFF && d[pos + 1] >= 0xD0 && d[pos + 1] <= 0xD7)) {
            if (d[pos] == 0xFF && d[pos + 1] != 0 && d[pos + 1] != 0xFF) return;     // some other marker: leave it to the caller
            pos++;
        }
        if (pos + 1 < n) pos += 2;
        eobrun = 0; for (Component &c : comp) c.pred = 0;
    }

    // ---- block decoders (T.81 F.2.2, G.1.2) ----
    // A DC difference category above 11 cannot come from 8-bit samples (T.81 Table F.1); a corrupt table that yields one would
    // turn into shifts by negative amounts below.
    void block_baseline(Component &c, int16_t *b, int td, int ta) {
        int s = decode(dc[td]);
        if (s > 11) { ok = false; return; }
        c.pred += extend(getbits(s), s);
        b[0] = (int16_t)c.pred;
        for (int k = 1; k < 64; ++k) {
            int rs = decode(ac[ta]), r = rs >> 4; s = rs & 15;
            if (s == 0) { if (r != 15) break; k += 15; continue; }
            k += r;
            b[kZigzag[k]] = (int16_t)extend(getbits(s), s);
        }
    }
    void block_dc_first(Component &c, int16_t *b, int td, int al) {
        int s = decode(dc[td]);
        if (s > 11) { ok = false; return; }
        c.pred += extend(getbits(s), s);
        b[0] = (int16_t)(c.pred * (1 << al));
    }
    void block_dc_refine(int16_t *b, int al) { if (getbit()) b[0] |= (int16_t)(1 << al); }
    void block_ac_first(int16_t *b, int ta, int ss, int se, int al) {
        if (eobrun > 0) { eobrun--; return; }
        for (int k = ss; k <= se; ++k) {
            int rs = decode(ac[ta]), r = rs >> 4, s = rs & 15;
            if (s == 0) {
                if (r != 15) { eobrun = (1 << r) - 1; if (r) eobrun += getbits(r); break; }
                k += 15; continue;
            }
            k += r;
            b[kZigzag[k]] = (int16_t)(extend(getbits(s), s) * (1 << al));
        }
    }
    void refine_nonzero(int16_t *p, int p1, int m1) {
        if (getbit() && (*p & p1) == 0) *p = (int16_t)(*p >= 0 ? *p + p1 : *p + m1);
    }
    void block_ac_refine(int16_t *b, int ta, int ss, int se, int al) {
        const int p1 = 1 << al, m1 = -(1 << al);
        int k = ss;
        if (eobrun == 0) {
            for (; k <= se; ++k) {
                int rs = decode(ac[ta]), r = rs >> 4, s = rs & 15;
                if (s) s = getbit() ? p1 : m1;
                else if (r != 15) { eobrun = 1 << r; if (r) eobrun += getbits(r); break; }
                do {
                    int16_t *p = b + kZigzag[k];
                    if (*p != 0) refine_nonzero(p, p1, m1);
                    else if (--r < 0) break;
                    k++;
                } while (k <= se);
                if (s && k <= se) b[kZigzag[k]] = (int16_t)s;
                if (!ok) return;
            }
        }
        if (eobrun > 0) {
            for (; k <= se; ++k) { int16_t *p = b + kZigzag[k]; if (*p != 0) refine_nonzero(p, p1, m1); }
            eobrun--;
        }
    }

    bool scan() {          // at the SOS segment's length field
        int len = u16(), ns = u8();
        if (ns < 1 || ns > ncomp || len != 6 + 2 * ns) return false;
        int ci[3], td[3], ta[3];
        for (int i = 0; i < ns; ++i) {
            int id = u8(), t = u8(); ci[i] = -1;
            for (int c = 0; c < ncomp; ++c) if (comp[c].id == id) ci[i] = c;
            if (ci[i] < 0) return false;
            td[i] = t >> 4; ta[i] = t & 15;
            if (td[i] > 3 || ta[i] > 3) return false;
        }
        int ss = u8(), se = u8(), a = u8(), ah = a >> 4, al = a & 15;
        if (!progressive) { ss = 0; se = 63; ah = al = 0; }
        if (ss > se || se > 63 || (ss == 0 && se != 0 && progressive) || (ss != 0 && ns != 1) || al > 13) return false;
        reset_entropy();
        auto one = [&](Component &c, int16_t *b, int i) {
            if (!progressive) block_baseline(c, b, td[i], ta[i]);
            else if (ss == 0) { if (ah == 0) block_dc_first(c, b, td[i], al); else block_dc_refine(b, al); }
            else if (ah == 0) block_ac_first(b, ta[i], ss, se, al);
            else block_ac_refine(b, ta[i], ss, se, al);
        };
        int count = 0;
        auto tick = [&]() { if (restart_interval && ++count == restart_interval) { count = 0; restart(); } };
        if (ns == 1) {     // non-interleaved: the component's own blocks, row by row
            Component &c = comp[ci[0]];
            int nbx = (c.w + 7) / 8, nby = (c.hgt + 7) / 8;
            for (int by = 0; by < nby && ok; ++by)
                for (int bx = 0; bx < nbx && ok; ++bx) { one(c, &c.coef[((size_t)by * c.bw + bx) * 64], 0); tick(); }
        } else {
            for (int my = 0; my < mcuy && ok; ++my)
                for (int mx = 0; mx < mcux && ok; ++mx) {
                    for (int i = 0; i < ns; ++i) {
                        Component &c = comp[ci[i]];
                        for (int y = 0; y < c.v; ++y)
                            for (int x = 0; x < c.h; ++x) one(c, &c.coef[((size_t)(my * c.v + y) * c.bw + (mx * c.h + x)) * 64], i);
                    }
                    tick();
                }
        }
        // skip what is left of the entropy-coded segment: up to the next marker that is not RSTn / stuffing
        while (pos + 1 < n && !(d[pos] == 0xFF && d[pos + 1] != 0 && d[pos + 1] != 0xFF && !(d[pos + 1] >= 0xD0 && d[pos + 1] <= 0xD7))) pos++;
        return ok;
    }

    bool headers_and_scans() {
        if (n < 4 || d[0] != 0xFF || d[1] != 0xD8) return false;
        pos = 2;
        bool have_frame = false, seen_scan = false;
        while (ok && pos + 4 <= n) {
            if (d[pos] != 0xFF) { pos++; continue; }
            int m = d[pos + 1];
            if (m == 0xFF) { pos++; continue; }
            pos += 2;
            if (m == 0xD9) break;
            if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;
            if (m == 0xDA) { if (!have_frame || !scan()) return false; seen_scan = true; continue; }
            size_t seg = pos; int len = u16();
            if (len < 2 || seg + len > n) return false;
            if (m == 0xC0 || m == 0xC1 || m == 0xC2) {
                if (have_frame) return false;
                progressive = m == 0xC2;
                int prec = u8(); H = u16(); W = u16(); ncomp = u8();
                if (prec != 8 || W <= 0 || H <= 0 || (ncomp != 1 && ncomp != 3)) return false;
                if ((size_t)W * (size_t)H > ((size_t)1 << 28)) return false;     // 268 M texels: beyond any texture, and the buffers below scale with it
                for (int c = 0; c < ncomp; ++c) {
                    comp[c].id = u8(); int hv = u8(); comp[c].h = hv >> 4; comp[c].v = hv & 15; comp[c].tq = u8();
                    if (comp[c].h < 1 || comp[c].h > 2 || comp[c].v < 1 || comp[c].v > 2 || comp[c].tq > 3) return false;
                    hmax = comp[c].h > hmax ? comp[c].h : hmax; vmax = comp[c].v > vmax ? comp[c].v : vmax;
                }
                if (ncomp == 1) { comp[0].h = comp[0].v = 1; hmax = vmax = 1; }
                mcux = (W + 8 * hmax - 1) / (8 * hmax); mcuy = (H + 8 * vmax - 1) / (8 * vmax);
                for (int c = 0; c < ncomp; ++c) {
                    Component &C = comp[c];
                    C.w = (W * C.h + hmax - 1) / hmax; C.hgt = (H * C.v + vmax - 1) / vmax;
                    C.bw = mcux * C.h; C.bh = mcuy * C.v;
                    C.coef.assign((size_t)C.bw * C.bh * 64, 0);
                }
                have_frame = true;
            } else if (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
                return false;                                      // lossless, hierarchical, arithmetic coding: not handled
            } else if (m == 0xC4) {
                while (ok && pos < seg + len) {
                    int t = u8(), cls = t >> 4, id = t & 15;
                    if (cls > 1 || id > 3) return false;
                    Huff &h = cls ? ac[id] : dc[id];
                    int total = 0;
                    for (int l = 1; l <= 16; ++l) { h.bits[l] = (uint8_t)u8(); total += h.bits[l]; }
                    if (total > 256 || pos + total > seg + len) return false;
                    for (int i = 0; i < total; ++i) h.vals[i] = (uint8_t)u8();
                    h.build();
                }
            } else if (m == 0xDB) {
                while (ok && pos < seg + len) {
                    int t = u8(), p16 = t >> 4, id = t & 15;
                    if (id > 3) return false;
                    for (int k = 0; k < 64; ++k) qt[id][kZigzag[k]] = (uint16_t)(p16 ? u16() : u8());
                    have_qt[id] = true;
                }
            } else if (m == 0xDD) restart_interval = u16();
            pos = seg + len;
        }
        return ok && have_frame && seen_scan;
    }

    // ---- inverse DCT (islow) ----
    static uint8_t clamp8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
    static void idct(const int16_t *in, const uint16_t *q, uint8_t *out, int stride) {
        const int CB = 13, P1 = 2;
        const long F0298 = 2446, F0390 = 3196, F0541 = 4433, F0765 = 6270, F0899 = 7373, F1175 = 9633, F1501 = 12299, F1847 = 15137, F1961 = 16069,
                   F2053 = 16819, F2562 = 20995, F3072 = 25172;
        long ws[64];
        for (int c = 0; c < 8; ++c) {
            long i0 = (long)in[c] * q[c], i1 = (long)in[8 + c] * q[8 + c], i2 = (long)in[16 + c] * q[16 + c], i3 = (long)in[24 + c] * q[24 + c],
                 i4 = (long)in[32 + c] * q[32 + c], i5 = (long)in[40 + c] * q[40 + c], i6 = (long)in[48 + c] * q[48 + c], i7 = (long)in[56 + c] * q[56 + c];
            long z2 = i2, z3 = i6;
            long z1 = (z2 + z3) * F0541, tmp2 = z1 + z3 * (-F1847), tmp3 = z1 + z2 * F0765;
            long tmp0 = (i0 + i4) * (1L << CB), tmp1 = (i0 - i4) * (1L << CB);
            long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
            tmp0 = i7; tmp1 = i5; tmp2 = i3; tmp3 = i1;
            z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2; long z4 = tmp1 + tmp3, z5 = (z3 + z4) * F1175;
            tmp0 *= F0298; tmp1 *= F2053; tmp2 *= F3072; tmp3 *= F1501;
            z1 *= -F0899; z2 *= -F2562; z3 *= -F1961; z4 *= -F0390;
            z3 += z5; z4 += z5;
            tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
            const int sh = CB - P1; const long rnd = 1L << (sh - 1);
            ws[c] = (tmp10 + tmp3 + rnd) >> sh; ws[56 + c] = (tmp10 - tmp3 + rnd) >> sh;
            ws[8 + c] = (tmp11 + tmp2 + rnd) >> sh; ws[48 + c] = (tmp11 - tmp2 + rnd) >> sh;
            ws[16 + c] = (tmp12 + tmp1 + rnd) >> sh; ws[40 + c] = (tmp12 - tmp1 + rnd) >> sh;
            ws[24 + c] = (tmp13 + tmp0 + rnd) >> sh; ws[32 + c] = (tmp13 - tmp0 + rnd) >> sh;
        }
        for (int r = 0; r < 8; ++r) {
            const long *w = ws + 8 * r;
            long z2 = w[2], z3 = w[6];
            long z1 = (z2 + z3) * F0541, tmp2 = z1 + z3 * (-F1847), tmp3 = z1 + z2 * F0765;
            long tmp0 = (w[0] + w[4]) * (1L << CB), tmp1 = (w[0] - w[4]) * (1L << CB);
            long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
            tmp0 = w[7]; tmp1 = w[5]; tmp2 = w[3]; tmp3 = w[1];
            z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2; long z4 = tmp1 + tmp3, z5 = (z3 + z4) * F1175;
            tmp0 *= F0298; tmp1 *= F2053; tmp2 *= F3072; tmp3 *= F1501;
            z1 *= -F0899; z2 *= -F2562; z3 *= -F1961; z4 *= -F0390;
            z3 += z5; z4 += z5;
            tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
            const int sh = CB + P1 + 3; const long rnd = 1L << (sh - 1);
            uint8_t *o = out + (size_t)r * stride;
            o[0] = clamp8((int)((tmp10 + tmp3 + rnd) >> sh) + 128); o[7] = clamp8((int)((tmp10 - tmp3 + rnd) >> sh) + 128);
            o[1] = clamp8((int)((tmp11 + tmp2 + rnd) >> sh) + 128); o[6] = clamp8((int)((tmp11 - tmp2 + rnd) >> sh) + 128);
            o[2] = clamp8((int)((tmp12 + tmp1 + rnd) >> sh) + 128); o[5] = clamp8((int)((tmp12 - tmp1 + rnd) >> sh) + 128);
            o[3] = clamp8((int)((tmp13 + tmp0 + rnd) >> sh) + 128); o[4] = clamp8((int)((tmp13 - tmp0 + rnd) >> sh) + 128);
        }
    }
    bool reconstruct() {
        for (int c = 0; c < ncomp; ++c) {
            Component &C = comp[c];
            if (!have_qt[C.tq]) return false;
            int stride = C.bw * 8;
            C.plane.assign((size_t)stride * C.bh * 8, 0);
            for (int by = 0; by < C.bh; ++by)
                for (int bx = 0; bx < C.bw; ++bx) idct(&C.coef[((size_t)by * C.bw + bx) * 64], qt[C.tq], &C.plane[(size_t)by * 8 * stride + bx * 8], stride);
            C.coef.clear(); C.coef.shrink_to_fit();
        }
        return true;
    }

    // ---- upsampling: one full-resolution row of component c into `out` (W samples) ----
    void upsample_row(const Component &C, int y, uint8_t *out) const {
        const int stride = C.bw * 8, cw = C.w;
        const bool h2 = C.h * 2 == hmax, v2 = C.v * 2 == vmax;
        auto rowp = [&](int r) { r = r < 0 ? 0 : (r >= C.hgt ? C.hgt - 1 : r); return &C.plane[(size_t)r * stride]; };
        if (!h2 && !v2) { memcpy(out, rowp(y), (size_t)W); return; }
        if (!h2) { memcpy(out, rowp(y / 2), (size_t)W); return; }                      // v2 only: row replication
        std::vector<uint8_t> tmp((size_t)2 * cw + 2);
        if (!v2) {                                                                     // h2v1 triangle filter
            const uint8_t *in = rowp(y);
            if (cw <= 2) { for (int x = 0; x < cw; ++x) tmp[2 * x] = tmp[2 * x + 1] = in[x]; }
            else {
                tmp[0] = in[0]; tmp[1] = (uint8_t)((in[0] * 3 + in[1] + 2) >> 2);
                for (int x = 1; x < cw - 1; ++x) { int v = in[x] * 3; tmp[2 * x] = (uint8_t)((v + in[x - 1] + 1) >> 2); tmp[2 * x + 1] = (uint8_t)((v + in[x + 1] + 2) >> 2); }
                int v = in[cw - 1] * 3; tmp[2 * cw - 2] = (uint8_t)((v + in[cw - 2] + 1) >> 2); tmp[2 * cw - 1] = in[cw - 1];
            }
        } else {                                                                       // h2v2 triangle filter
            const int cy = y / 2;
            const uint8_t *in0 = rowp(cy), *in1 = rowp((y & 1) ? cy + 1 : cy - 1);
            if (cw <= 2) { for (int x = 0; x < cw; ++x) tmp[2 * x] = tmp[2 * x + 1] = in0[x]; }
            else {
                int thiscol = in0[0] * 3 + in1[0], nextcol = in0[1] * 3 + in1[1], lastcol;
                tmp[0] = (uint8_t)((thiscol * 4 + 8) >> 4); tmp[1] = (uint8_t)((thiscol * 3 + nextcol + 7) >> 4);
                lastcol = thiscol; thiscol = nextcol;
                for (int x = 1; x < cw - 1; ++x) {
                    nextcol = in0[x + 1] * 3 + in1[x + 1];
                    tmp[2 * x] = (uint8_t)((thiscol * 3 + lastcol + 8) >> 4); tmp[2 * x + 1] = (uint8_t)((thiscol * 3 + nextcol + 7) >> 4);
                    lastcol = thiscol; thiscol = nextcol;
                }
                tmp[2 * cw - 2] = (uint8_t)((thiscol * 3 + lastcol + 8) >> 4); tmp[2 * cw - 1] = (uint8_t)((thiscol * 4 + 7) >> 4);
            }
        }
        memcpy(out, tmp.data(), (size_t)W);
    }

    bool to_rgb(std::vector<uint8_t> *rgb) const {
        rgb->assign((size_t)W * H * 3, 0);
        if (ncomp == 1) {
            const int stride = comp[0].bw * 8;
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x) { uint8_t v = comp[0].plane[(size_t)y * stride + x]; uint8_t *o = &(*rgb)[((size_t)y * W + x) * 3]; o[0] = o[1] = o[2] = v; }
            return true;
        }
        int crr[256], cbb[256]; long crg[256], cbg[256];
        for (int i = 0; i < 256; ++i) {
            long x = i - 128;
            crr[i] = (int)((91881L * x + 32768L) >> 16); cbb[i] = (int)((116130L * x + 32768L) >> 16);
            crg[i] = -46802L * x; cbg[i] = -22554L * x + 32768L;
        }
        std::vector<uint8_t> ry((size_t)W + 16), rcb((size_t)W + 16), rcr((size_t)W + 16);
        for (int y = 0; y < H; ++y) {
            upsample_row(comp[0], y, ry.data()); upsample_row(comp[1], y, rcb.data()); upsample_row(comp[2], y, rcr.data());
            uint8_t *o = &(*rgb)[(size_t)y * W * 3];
            for (int x = 0; x < W; ++x) {
                int Y = ry[x], cb = rcb[x], cr = rcr[x];
                o[3 * x] = clamp8(Y + crr[cr]); o[3 * x + 1] = clamp8(Y + (int)((cbg[cb] + crg[cr]) >> 16)); o[3 * x + 2] = clamp8(Y + cbb[cb]);
            }
        }
        return true;
    }
};

}  // namespace

bool DecodeJPEG(const uint8_t *data, size_t size, int *width, int *height, std::vector<uint8_t> *rgb) {
    if (!data || !width || !height || !rgb) return false;
    try {
        Decoder dec(data, size);
        if (!dec.headers_and_scans() || !dec.reconstruct() || !dec.to_rgb(rgb)) return false;
        *width = dec.W; *height = dec.H;
    } catch (std::exception &) { return false; }      // allocation failure on a hostile header
    return true;
}
