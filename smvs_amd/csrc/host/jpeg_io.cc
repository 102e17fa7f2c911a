// JPEG decoder (sequential and progressive) for the byte-image embeddings of an MVE view
// ("original.jpg": smvsrecon --image=original, app/smvsrecon.cc:41, 156;
// makescene keeps the camera's JPEG as the `original` embedding).  The
// reference reads them through mve::image::load_jpg_file, i.e. libjpeg with its
// default decompression parameters [MVE-unverified M32]: the accurate integer
// inverse DCT (JDCT_ISLOW, jidctint.c), "fancy" (triangle) chroma upsampling
// for 2:1 factors (jdsample.c), the 16-bit fixed-point YCbCr -> RGB tables
// (jdcolor.c).  This file restates those published algorithms so that the
// pixels are the ones libjpeg produces: tests/test_scene_io_cpu.py compares
// with Pillow's decoder (libjpeg-turbo, bit-identical to libjpeg for these
// methods) on 4:4:4 / 4:2:2 / 4:2:0 / 4:1:1 / grey images, sequential and
// progressive, optimised Huffman tables, restart intervals, sizes that are no
// multiple of the MCU.
//
// Supported: SOF0 / SOF1 (sequential) and SOF2 (progressive: spectral
// selection and successive approximation, jdphuff.c), Huffman, 8 bits, one or
// three components, sampling factors 1 .. 4, interleaved and non-interleaved
// scans.  Every scan decodes into the frame's coefficient arrays; the inverse
// DCT runs once at the end (for a complete progressive file libjpeg's block
// smoothing is inactive, so the pixels are again libjpeg's).  Refused with a
// reason: arithmetic-coded, lossless and 12-bit files, CMYK / YCCK.
#include "jpeg_io.h"

#include <array>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <vector>

namespace smvs_amd {
namespace {

typedef std::vector<unsigned char> Bytes;

Bytes
read_file(std::string const& path)
{
    std::ifstream in(path.c_str(), std::ios::binary);
    if (!in)
        throw std::runtime_error("cannot open " + path);
    in.seekg(0, std::ios::end);
    std::streamoff const size = in.tellg();
    in.seekg(0, std::ios::beg);
    Bytes data((std::size_t)size);
    in.read(reinterpret_cast<char*>(data.data()), size);
    if (!in)
        throw std::runtime_error("cannot read " + path);
    return data;
}

int const ZIGZAG[64] = {   // jutils.c: jpeg_natural_order
    0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5,
    12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
    58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63 };

struct HuffTable {
    bool present = false;
    // jdhuff.c, jpeg_make_d_derived_tbl: canonical codes by length
    int maxcode[18];       // largest code of length l (-1: none)
    int valoffset[17];     // huffval[] index of the first code of length l
    unsigned char huffval[256];
    int count = 0;
    // codes of <= LOOK bits by their first LOOK bits: length << 8 | symbol (0: longer)
    static constexpr int LOOK = 9;
    uint16_t look[1 << LOOK];
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0;
    int td = 0, ta = 0;                 // Huffman table selectors of the scan
    int width_blocks = 0, height_blocks = 0;   // padded to whole MCUs
    int down_w = 0, down_h = 0;         // downsampled_width / _height
    std::vector<unsigned char> plane;   // [height_blocks * 8][width_blocks * 8]
    std::vector<int16_t> coefs;         // [height_blocks][width_blocks][64], natural order
    uint16_t q[64];                     // its quantisation table, natural order, latched
    bool q_latched = false;             //   at the component's first scan (jdinput.c)
    int pred = 0;
};

struct Scan {
    int ns = 0;
    Component* comp[4] = { nullptr, nullptr, nullptr, nullptr };
    int ss = 0, se = 63, ah = 0, al = 0;
};

struct Decoder {
    std::string path;
    Bytes data;
    std::size_t pos = 0;
    int width = 0, height = 0;
    std::vector<Component> comps;
    uint16_t quant[4][64];
    bool quant_present[4] = { false, false, false, false };
    HuffTable dc[4], ac[4];
    int restart_interval = 0;
    bool saw_jfif = false, saw_adobe = false;
    int adobe_transform = 0;
    bool have_frame = false, progressive = false;
    int eobrun = 0;
    // bit reader
    uint32_t bitbuf = 0;
    int bits = 0;
    bool hit_marker = false;

    [[noreturn]] void fail(std::string const& what) const
    {
        throw std::runtime_error("JPEG " + path + ": " + what);
    }
    int u8(void)
    {
        if (pos >= data.size())
            fail("truncated file");
        return data[pos++];
    }
    int u16(void)
    {
        int const a = u8();
        return (a << 8) | u8();
    }
};

// post-IDCT range limit (jdmaster.c, prepare_range_limit_table): the sample
// for the descaled value v is table[v & 1023] with 0..127 -> 128 + v,
// 128..511 -> 255, 512..895 -> 0, 896..1023 -> v - 896
inline unsigned char
idct_limit(int v)
{
    int const i = v & 1023;
    if (i < 128)
        return (unsigned char)(128 + i);
    if (i < 512)
        return 255;
    if (i < 896)
        return 0;
    return (unsigned char)(i - 896);
}

inline unsigned char
clamp_u8(int v)
{
    return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// jidctint.c: jpeg_idct_islow (CONST_BITS 13, PASS1_BITS 2), with its
// shortcuts for columns / rows whose AC terms are zero (they round differently
// from the general path, so they are part of the result)
void
idct_islow(const int* coef /*[64], dequantised*/, unsigned char* out, std::size_t stride)
{
    constexpr int CB = 13, P1 = 2;
    constexpr long F_0_298631336 = 2446, F_0_390180644 = 3196, F_0_541196100 = 4433,
        F_0_765366865 = 6270, F_0_899976223 = 7373, F_1_175875602 = 9633,
        F_1_501321110 = 12299, F_1_847759065 = 15137, F_1_961570560 = 16069,
        F_2_053119869 = 16819, F_2_562915447 = 20995, F_3_072711026 = 25172;
    auto descale = [](long x, int n) -> long { return (x + (1L << (n - 1))) >> n; };
    long ws[64];
    for (int c = 0; c < 8; ++c) {
        const int* in = coef + c;
        long* w = ws + c;
        if (in[8] == 0 && in[16] == 0 && in[24] == 0 && in[32] == 0 && in[40] == 0
            && in[48] == 0 && in[56] == 0) {
            long const dcval = (long)in[0] * (1L << P1);
            for (int r = 0; r < 8; ++r)
                w[8 * r] = dcval;
            continue;
        }
        long z2 = in[16], z3 = in[48];
        long z1 = (z2 + z3) * F_0_541196100;
        long tmp2 = z1 + z3 * (-F_1_847759065);
        long tmp3 = z1 + z2 * F_0_765366865;
        z2 = in[0]; z3 = in[32];
        long tmp0 = (z2 + z3) * (1L << CB);
        long tmp1 = (z2 - z3) * (1L << CB);
        long const tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3;
        long const tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = in[56]; tmp1 = in[40]; tmp2 = in[24]; tmp3 = in[8];
        z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
        long z4 = tmp1 + tmp3;
        long const z5 = (z3 + z4) * F_1_175875602;
        tmp0 *= F_0_298631336; tmp1 *= F_2_053119869;
        tmp2 *= F_3_072711026; tmp3 *= F_1_501321110;
        z1 *= -F_0_899976223; z2 *= -F_2_562915447;
        z3 *= -F_1_961570560; z4 *= -F_0_390180644;
        z3 += z5; z4 += z5;
        tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
        w[0] = descale(tmp10 + tmp3, CB - P1);
        w[56] = descale(tmp10 - tmp3, CB - P1);
        w[8] = descale(tmp11 + tmp2, CB - P1);
        w[48] = descale(tmp11 - tmp2, CB - P1);
        w[16] = descale(tmp12 + tmp1, CB - P1);
        w[40] = descale(tmp12 - tmp1, CB - P1);
        w[24] = descale(tmp13 + tmp0, CB - P1);
        w[32] = descale(tmp13 - tmp0, CB - P1);
    }
    for (int r = 0; r < 8; ++r) {
        const long* w = ws + 8 * r;
        unsigned char* o = out + (std::size_t)r * stride;
        if (w[1] == 0 && w[2] == 0 && w[3] == 0 && w[4] == 0 && w[5] == 0 && w[6] == 0
            && w[7] == 0) {
            unsigned char const dcval = idct_limit((int)descale(w[0], P1 + 3));
            for (int c = 0; c < 8; ++c)
                o[c] = dcval;
            continue;
        }
        long z2 = w[2], z3 = w[6];
        long z1 = (z2 + z3) * F_0_541196100;
        long tmp2 = z1 + z3 * (-F_1_847759065);
        long tmp3 = z1 + z2 * F_0_765366865;
        long tmp0 = (w[0] + w[4]) * (1L << CB);
        long tmp1 = (w[0] - w[4]) * (1L << CB);
        long const tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3;
        long const tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = w[7]; tmp1 = w[5]; tmp2 = w[3]; tmp3 = w[1];
        z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
        long z4 = tmp1 + tmp3;
        long const z5 = (z3 + z4) * F_1_175875602;
        tmp0 *= F_0_298631336; tmp1 *= F_2_053119869;
        tmp2 *= F_3_072711026; tmp3 *= F_1_501321110;
        z1 *= -F_0_899976223; z2 *= -F_2_562915447;
        z3 *= -F_1_961570560; z4 *= -F_0_390180644;
        z3 += z5; z4 += z5;
        tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
        constexpr int S = CB + P1 + 3;
        o[0] = idct_limit((int)descale(tmp10 + tmp3, S));
        o[7] = idct_limit((int)descale(tmp10 - tmp3, S));
        o[1] = idct_limit((int)descale(tmp11 + tmp2, S));
        o[6] = idct_limit((int)descale(tmp11 - tmp2, S));
        o[2] = idct_limit((int)descale(tmp12 + tmp1, S));
        o[5] = idct_limit((int)descale(tmp12 - tmp1, S));
        o[3] = idct_limit((int)descale(tmp13 + tmp0, S));
        o[4] = idct_limit((int)descale(tmp13 - tmp0, S));
    }
}

void
build_huffman(Decoder& d, HuffTable& t, const unsigned char* bits /*[17], [0] unused*/)
{
    // jdhuff.c: codes of each length in increasing order
    int code = 0, k = 0;
    for (int l = 1; l <= 16; ++l) {
        t.valoffset[l] = k - code;
        if (bits[l] > 0) {
            k += bits[l];
            code += bits[l];
            t.maxcode[l] = code - 1;
        } else {
            t.maxcode[l] = -1;
        }
        if (code > (1 << l))
            d.fail("bad Huffman table");
        code <<= 1;
    }
    t.maxcode[17] = 0x7FFFFFFF;
    t.count = k;
    t.present = true;
    std::memset(t.look, 0, sizeof(t.look));
    int c = 0, idx = 0;
    for (int l = 1; l <= 16; ++l) {
        for (int i = 0; i < bits[l]; ++i, ++c, ++idx)
            if (l <= HuffTable::LOOK) {
                int const first = c << (HuffTable::LOOK - l);
                for (int f = 0; f < (1 << (HuffTable::LOOK - l)); ++f)
                    t.look[first + f] = (uint16_t)((l << 8) | t.huffval[idx]);
            }
        c <<= 1;
    }
}

// ---- entropy-coded segment: bits, MSB first, FF00 -> FF, markers end it ----
inline void
fill_bits(Decoder& d, int need)
{
    while (d.bits < need) {
        int byte = 0;
        if (!d.hit_marker && d.pos < d.data.size()) {
            byte = d.data[d.pos];
            if (byte == 0xFF) {
                int const next = d.pos + 1 < d.data.size() ? d.data[d.pos + 1] : 0xD9;
                if (next == 0x00) {
                    d.pos += 2;
                } else {
                    // a marker: the decoder sees zero bits from here on (jdhuff.c
                    // does the same and warns if they are really used)
                    d.hit_marker = true;
                    byte = 0;
                }
            } else {
                d.pos += 1;
            }
        } else if (!d.hit_marker) {
            // the data ends without a marker: a truncated file, not an image
            d.fail("truncated file (entropy-coded data ends without a marker)");
        }
        d.bitbuf = (d.bitbuf << 8) | (uint32_t)byte;
        d.bits += 8;
    }
}

inline int
get_bits(Decoder& d, int n)
{
    if (n == 0)
        return 0;
    fill_bits(d, n);
    d.bits -= n;
    return (int)((d.bitbuf >> d.bits) & ((1u << n) - 1u));
}

inline int
decode_symbol(Decoder& d, HuffTable const& t)
{
    // (the short codes -- nearly all of them -- through a table)
    fill_bits(d, HuffTable::LOOK);
    uint16_t const hit = t.look[(d.bitbuf >> (d.bits - HuffTable::LOOK))
        & ((1u << HuffTable::LOOK) - 1u)];
    if (hit != 0) {
        d.bits -= hit >> 8;
        return hit & 0xFF;
    }
    int code = get_bits(d, 1);
    int l = 1;
    while (l <= 16 && code > t.maxcode[l]) {
        code = (code << 1) | get_bits(d, 1);
        l += 1;
    }
    if (l > 16)
        d.fail("corrupt entropy-coded data (bad Huffman code)");
    int const idx = code + t.valoffset[l];
    if (idx < 0 || idx >= t.count)
        d.fail("corrupt entropy-coded data (bad Huffman code)");
    return t.huffval[idx];
}

inline int
extend(int v, int s)   // HUFF_EXTEND
{
    return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v;
}

// ---- one block of a scan into the component's coefficient array ----
// (coefficients in natural order, not yet dequantised)

// sequential: DC difference and the AC run lengths of the whole block (jdhuff.c)
void
block_sequential(Decoder& d, Component& c, int16_t* coef)
{
    HuffTable const& dct = d.dc[c.td];
    HuffTable const& act = d.ac[c.ta];
    int const s = decode_symbol(d, dct);
    if (s > 15)
        d.fail("corrupt entropy-coded data (DC size)");
    c.pred += s ? extend(get_bits(d, s), s) : 0;
    if (c.pred < -(1 << 20) || c.pred > (1 << 20))
        d.fail("corrupt entropy-coded data (DC value out of range)");
    coef[0] = (int16_t)c.pred;
    for (int k = 1; k < 64;) {
        int const rs = decode_symbol(d, act);
        int const r = rs >> 4, sz = rs & 15;
        if (sz == 0) {
            if (r != 15)
                break;          // EOB
            k += 16;            // ZRL
            continue;
        }
        k += r;
        if (k > 63)
            d.fail("corrupt entropy-coded data (run past the block)");
        coef[ZIGZAG[k]] = (int16_t)extend(get_bits(d, sz), sz);
        k += 1;
    }
}

// progressive, jdphuff.c: decode_mcu_DC_first / _DC_refine / _AC_first / _AC_refine
void
block_dc_first(Decoder& d, Component& c, int16_t* coef, int al)
{
    int const s = decode_symbol(d, d.dc[c.td]);
    if (s > 15)
        d.fail("corrupt entropy-coded data (DC size)");
    c.pred += s ? extend(get_bits(d, s), s) : 0;
    if (c.pred < -(1 << 20) || c.pred > (1 << 20))
        d.fail("corrupt entropy-coded data (DC value out of range)");
    coef[0] = (int16_t)(c.pred * (1 << al));
}

void
block_dc_refine(Decoder& d, int16_t* coef, int al)
{
    if (get_bits(d, 1))
        coef[0] = (int16_t)(coef[0] | (1 << al));
}

void
block_ac_first(Decoder& d, Component& c, int16_t* coef, Scan const& sc)
{
    if (d.eobrun > 0) {
        d.eobrun -= 1;
        return;
    }
    HuffTable const& act = d.ac[c.ta];
    for (int k = sc.ss; k <= sc.se; ++k) {
        int const rs = decode_symbol(d, act);
        int const r = rs >> 4, sz = rs & 15;
        if (sz != 0) {
            k += r;
            if (k > 63)
                d.fail("corrupt entropy-coded data (run past the block)");
            coef[ZIGZAG[k]] = (int16_t)(extend(get_bits(d, sz), sz) * (1 << sc.al));
        } else if (r == 15) {
            k += 15;            // ZRL
        } else {
            d.eobrun = 1 << r;  // EOBr
            if (r)
                d.eobrun += get_bits(d, r);
            d.eobrun -= 1;      // (this band is one of them)
            break;
        }
    }
}

void
block_ac_refine(Decoder& d, Component& c, int16_t* coef, Scan const& sc)
{
    int const p1 = 1 << sc.al, m1 = -(1 << sc.al);
    HuffTable const& act = d.ac[c.ta];
    auto correct = [&](int16_t* v) {
        // a correction bit for a coefficient that is already non-zero
        if (get_bits(d, 1) && (*v & p1) == 0)
            *v = (int16_t)(*v >= 0 ? *v + p1 : *v + m1);
    };
    int k = sc.ss;
    if (d.eobrun == 0) {
        for (; k <= sc.se; ++k) {
            int const rs = decode_symbol(d, act);
            int r = rs >> 4, sz = rs & 15;
            int value = 0;
            if (sz != 0) {
                if (sz != 1)
                    d.fail("corrupt entropy-coded data (refinement size)");
                value = get_bits(d, 1) ? p1 : m1;
            } else if (r != 15) {
                d.eobrun = 1 << r;
                if (r)
                    d.eobrun += get_bits(d, r);
                break;          // (the rest of the band is handled below)
            }
            // skip r zero coefficients, correcting the non-zero ones on the way
            for (; k <= sc.se; ++k) {
                int16_t* v = coef + ZIGZAG[k];
                if (*v != 0)
                    correct(v);
                else if (--r < 0)
                    break;
            }
            if (value != 0) {
                if (k > 63)
                    d.fail("corrupt entropy-coded data (run past the block)");
                coef[ZIGZAG[k]] = (int16_t)value;
            }
        }
    }
    if (d.eobrun > 0) {
        for (; k <= sc.se; ++k) {
            int16_t* v = coef + ZIGZAG[k];
            if (*v != 0)
                correct(v);
        }
        d.eobrun -= 1;
    }
}

// Tables and headers up to the next scan (true) or the end of the image (false).
bool
next_scan(Decoder& d, Scan* sc)
{
    for (;;) {
        // (between segments: fill bytes and, after a scan, whatever the entropy
        // decoder left unread in front of the marker)
        int b = d.u8();
        if (b != 0xFF)
            continue;
        while ((b = d.u8()) == 0xFF) {}
        int const marker = b;
        if (marker == 0x00 || marker == 0x01 || (marker >= 0xD0 && marker <= 0xD7))
            continue;
        if (marker == 0xD8)
            continue;           // SOI
        if (marker == 0xD9)
            return false;       // EOI
        int const len = d.u16();
        if (len < 2 || d.pos + (std::size_t)len - 2 > d.data.size())
            d.fail("bad segment length");
        std::size_t const end = d.pos + (std::size_t)len - 2;
        if (marker == 0xDB) {           // DQT
            while (d.pos < end) {
                int const pq = d.u8();
                int const prec = pq >> 4, id = pq & 15;
                if (id > 3)
                    d.fail("bad quantisation table id");
                for (int i = 0; i < 64; ++i)    // (stored in zigzag order)
                    d.quant[id][ZIGZAG[i]] = (uint16_t)(prec ? d.u16() : d.u8());
                d.quant_present[id] = true;
            }
        } else if (marker == 0xC4) {    // DHT
            while (d.pos < end) {
                int const tc = d.u8();
                int const cls = tc >> 4, id = tc & 15;
                if (cls > 1 || id > 3)
                    d.fail("bad Huffman table id");
                unsigned char bits[17] = { 0 };
                int total = 0;
                for (int l = 1; l <= 16; ++l) {
                    bits[l] = (unsigned char)d.u8();
                    total += bits[l];
                }
                if (total > 256)
                    d.fail("bad Huffman table");
                HuffTable& t = cls ? d.ac[id] : d.dc[id];
                for (int i = 0; i < total; ++i)
                    t.huffval[i] = (unsigned char)d.u8();
                build_huffman(d, t, bits);
            }
        } else if (marker == 0xDD) {    // DRI
            d.restart_interval = d.u16();
        } else if (marker == 0xE0) {    // APP0
            if (len >= 7 && std::memcmp(&d.data[d.pos], "JFIF\0", 5) == 0)
                d.saw_jfif = true;
        } else if (marker == 0xEE) {    // APP14
            if (len >= 14 && std::memcmp(&d.data[d.pos], "Adobe", 5) == 0) {
                d.saw_adobe = true;
                d.adobe_transform = d.data[d.pos + 11];
            }
        } else if (marker == 0xC0 || marker == 0xC1 || marker == 0xC2) {   // SOF0 / 1 / 2
            if (d.have_frame)
                d.fail("two frame headers");
            d.progressive = marker == 0xC2;
            int const precision = d.u8();
            d.height = d.u16();
            d.width = d.u16();
            int const n = d.u8();
            if (precision != 8)
                d.fail(std::to_string(precision) + "-bit samples are not byte images");
            if (n == 4)
                d.fail("CMYK / YCCK files are not supported");
            if (n != 1 && n != 3)
                d.fail("unsupported number of components");
            // (untrusted header: bound the size before anything is allocated)
            if (d.width <= 0 || d.height <= 0 || d.width > (1 << 16) || d.height > (1 << 16)
                || (long long)d.width * d.height * n > (1ll << 30))
                d.fail("image dimensions out of range");
            d.comps.resize((std::size_t)n);
            for (Component& c : d.comps) {
                c.id = d.u8();
                int const hv = d.u8();
                c.h = hv >> 4;
                c.v = hv & 15;
                c.tq = d.u8();
                if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.tq > 3)
                    d.fail("bad component description");
            }
            d.have_frame = true;
        } else if ((marker >= 0xC3 && marker <= 0xCF) && marker != 0xC4 && marker != 0xC8
            && marker != 0xCC) {
            d.fail("lossless / hierarchical / arithmetic-coded JPEG is not supported");
        } else if (marker == 0xCC) {
            d.fail("arithmetic-coded JPEG is not supported");
        } else if (marker == 0xDA) {    // SOS
            if (!d.have_frame)
                d.fail("scan before the frame header");
            sc->ns = d.u8();
            if (sc->ns < 1 || sc->ns > (int)d.comps.size())
                d.fail("bad number of components in a scan");
            for (int i = 0; i < sc->ns; ++i) {
                int const id = d.u8();
                int const tables = d.u8();
                Component* c = nullptr;
                for (Component& k : d.comps)
                    if (k.id == id)
                        c = &k;
                if (c == nullptr || (i > 0 && c <= sc->comp[i - 1]))
                    d.fail("scan components out of order");
                c->td = tables >> 4;
                c->ta = tables & 15;
                if (c->td > 3 || c->ta > 3)
                    d.fail("bad Huffman table selector");
                sc->comp[i] = c;
            }
            sc->ss = d.u8();
            sc->se = d.u8();
            int const a = d.u8();
            sc->ah = a >> 4;
            sc->al = a & 15;
            if (!d.progressive) {
                // B.2.3: a sequential scan covers the whole block
                sc->ss = 0; sc->se = 63; sc->ah = 0; sc->al = 0;
            } else if (sc->ss > sc->se || sc->se > 63 || sc->ah > 13 || sc->al > 13
                || (sc->ss == 0 && sc->se != 0) || (sc->ss > 0 && sc->ns != 1)) {
                d.fail("bad progression parameters");
            }
            for (int i = 0; i < sc->ns; ++i) {
                Component* c = sc->comp[i];
                bool const need_dc = sc->ss == 0 && sc->ah == 0;
                bool const need_ac = sc->se > 0;
                if ((need_dc && !d.dc[c->td].present) || (need_ac && !d.ac[c->ta].present)
                    || !d.quant_present[c->tq])
                    d.fail("scan refers to a table that was not defined");
                if (!c->q_latched) {
                    std::memcpy(c->q, d.quant[c->tq], sizeof(c->q));
                    c->q_latched = true;
                }
            }
            d.pos = end;
            return true;
        }
        d.pos = end;
    }
}

// The entropy-coded data of one scan (A.2.3 / B.2.3 for the MCU order).
void
decode_scan(Decoder& d, Scan const& sc, int mcux, int mcuy)
{
    bool const interleaved = sc.ns > 1;
    int const nx = interleaved ? mcux : (sc.comp[0]->down_w + 7) / 8;
    int const ny = interleaved ? mcuy : (sc.comp[0]->down_h + 7) / 8;
    d.bits = 0;
    d.bitbuf = 0;
    d.hit_marker = false;
    d.eobrun = 0;
    for (int i = 0; i < sc.ns; ++i)
        sc.comp[i]->pred = 0;
    int restarts_left = d.restart_interval;
    int next_rst = 0;
    for (int my = 0; my < ny; ++my)
        for (int mx = 0; mx < nx; ++mx) {
            if (d.restart_interval > 0 && restarts_left == 0) {
                // byte-align, expect RSTn
                d.bits = 0;
                d.bitbuf = 0;
                d.hit_marker = false;
                while (d.pos + 1 < d.data.size()
                    && !(d.data[d.pos] == 0xFF && d.data[d.pos + 1] >= 0xD0
                        && d.data[d.pos + 1] <= 0xD7)) {
                    if (d.data[d.pos] == 0xFF && d.data[d.pos + 1] != 0x00
                        && d.data[d.pos + 1] != 0xFF)
                        d.fail("restart marker expected");
                    d.pos += 1;
                }
                if (d.pos + 1 >= d.data.size() || d.data[d.pos + 1] != 0xD0 + next_rst)
                    d.fail("restart markers out of sequence");
                d.pos += 2;
                next_rst = (next_rst + 1) & 7;
                restarts_left = d.restart_interval;
                d.eobrun = 0;
                for (int i = 0; i < sc.ns; ++i)
                    sc.comp[i]->pred = 0;
            }
            for (int i = 0; i < sc.ns; ++i) {
                Component& c = *sc.comp[i];
                int const bh = interleaved ? c.h : 1, bv = interleaved ? c.v : 1;
                for (int by = 0; by < bv; ++by)
                    for (int bx = 0; bx < bh; ++bx) {
                        int16_t* coef = c.coefs.data()
                            + ((std::size_t)(my * bv + by) * c.width_blocks + (mx * bh + bx)) * 64;
                        if (!d.progressive)
                            block_sequential(d, c, coef);
                        else if (sc.ss == 0 && sc.ah == 0)
                            block_dc_first(d, c, coef, sc.al);
                        else if (sc.ss == 0)
                            block_dc_refine(d, coef, sc.al);
                        else if (sc.ah == 0)
                            block_ac_first(d, c, coef, sc);
                        else
                            block_ac_refine(d, c, coef, sc);
                    }
            }
            if (d.restart_interval > 0)
                restarts_left -= 1;
        }
    // (the bits of the last byte are padding; the next marker follows)
    d.bits = 0;
    d.bitbuf = 0;
    d.hit_marker = false;
}

// ---- chroma upsampling (jdsample.c) of one component to full resolution ----
// in: the component's plane (stride = width_blocks * 8), of which down_w x
// down_h samples are real; out: [H][W] with W = down_w * hf rounded ... the
// caller crops to the image.
void
upsample(Component const& c, int hf, int vf, int out_w, int out_h,
    std::vector<unsigned char>* out)
{
    std::size_t const stride = (std::size_t)c.width_blocks * 8;
    int const dw = c.down_w, dh = c.down_h;
    std::size_t const ow = (std::size_t)dw * hf;
    out->assign(ow * ((std::size_t)dh * vf), 0);
    auto row_in = [&](int r) {
        // context rows beyond the component are its edge rows (jdmainct.c)
        r = r < 0 ? 0 : (r >= dh ? dh - 1 : r);
        return c.plane.data() + (std::size_t)r * stride;
    };
    if (hf == 1 && vf == 1) {
        for (int r = 0; r < dh; ++r)
            std::memcpy(out->data() + (std::size_t)r * ow, row_in(r), (std::size_t)dw);
    } else if (hf == 2 && vf == 1) {
        // h2v1_fancy_upsample: 3/4 nearer + 1/4 further, biases 1 and 2
        for (int r = 0; r < dh; ++r) {
            const unsigned char* in = row_in(r);
            unsigned char* o = out->data() + (std::size_t)r * ow;
            if (dw <= 2) {
                // (jdsample.c: the triangle filter only when downsampled_width > 2,
                // h2v1_upsample's replication otherwise)
                for (int i = 0; i < dw; ++i)
                    o[2 * i] = o[2 * i + 1] = in[i];
                continue;
            }
            o[0] = in[0];
            o[1] = (unsigned char)((in[0] * 3 + in[1] + 2) >> 2);
            for (int i = 1; i < dw - 1; ++i) {
                int const v = in[i] * 3;
                o[2 * i] = (unsigned char)((v + in[i - 1] + 1) >> 2);
                o[2 * i + 1] = (unsigned char)((v + in[i + 1] + 2) >> 2);
            }
            o[2 * dw - 2] = (unsigned char)((in[dw - 1] * 3 + in[dw - 2] + 1) >> 2);
            o[2 * dw - 1] = in[dw - 1];
        }
    } else if (hf == 2 && vf == 2) {
        // h2v2_fancy_upsample: 9/16, 3/16, 3/16, 1/16; biases 8 and 7
        for (int r = 0; r < dh; ++r)
            for (int half = 0; half < 2; ++half) {
                const unsigned char* in0 = row_in(r);
                const unsigned char* in1 = row_in(half == 0 ? r - 1 : r + 1);
                unsigned char* o = out->data() + ((std::size_t)2 * r + half) * ow;
                if (dw <= 2) {
                    // (jdsample.c: h2v2_upsample's replication unless
                    // downsampled_width > 2)
                    for (int i = 0; i < dw; ++i)
                        o[2 * i] = o[2 * i + 1] = in0[i];
                    continue;
                }
                int thiscol = in0[0] * 3 + in1[0];
                int nextcol = in0[1] * 3 + in1[1];
                o[0] = (unsigned char)((thiscol * 4 + 8) >> 4);
                o[1] = (unsigned char)((thiscol * 3 + nextcol + 7) >> 4);
                int lastcol = thiscol;
                thiscol = nextcol;
                for (int i = 1; i < dw - 1; ++i) {
                    nextcol = in0[i + 1] * 3 + in1[i + 1];
                    o[2 * i] = (unsigned char)((thiscol * 3 + lastcol + 8) >> 4);
                    o[2 * i + 1] = (unsigned char)((thiscol * 3 + nextcol + 7) >> 4);
                    lastcol = thiscol;
                    thiscol = nextcol;
                }
                o[2 * dw - 2] = (unsigned char)((thiscol * 3 + lastcol + 8) >> 4);
                o[2 * dw - 1] = (unsigned char)((thiscol * 4 + 7) >> 4);
            }
    } else if (hf == 1 && vf == 2) {
        // h1v2_fancy_upsample (libjpeg-turbo): 3/4 nearer row + 1/4 further,
        // bias 1 for the upper, 2 for the lower output row
        for (int r = 0; r < dh; ++r)
            for (int half = 0; half < 2; ++half) {
                const unsigned char* in0 = row_in(r);
                const unsigned char* in1 = row_in(half == 0 ? r - 1 : r + 1);
                unsigned char* o = out->data() + ((std::size_t)2 * r + half) * ow;
                int const bias = half == 0 ? 1 : 2;
                for (int i = 0; i < dw; ++i)
                    o[i] = (unsigned char)((in0[i] * 3 + in1[i] + bias) >> 2);
            }
    } else {
        // int_upsample: replication
        for (int r = 0; r < dh * vf; ++r) {
            const unsigned char* in = row_in(r / vf);
            unsigned char* o = out->data() + (std::size_t)r * ow;
            for (int i = 0; i < dw; ++i)
                for (int k = 0; k < hf; ++k)
                    o[(std::size_t)i * hf + k] = in[i];
        }
    }
    (void)out_w;
    (void)out_h;
}

ByteImage::Ptr
decode(std::string const& path, bool header_only, int* whc)
{
    Decoder d;
    d.path = path;
    d.data = read_file(path);
    if (d.u8() != 0xFF || d.u8() != 0xD8)
        d.fail("not a JPEG file (no SOI)");
    Scan sc;
    if (!next_scan(d, &sc))
        d.fail("no image data (EOI)");
    int const n = (int)d.comps.size();
    if (whc != nullptr) {
        whc[0] = d.width;
        whc[1] = d.height;
        whc[2] = n;
    }
    if (header_only)
        return ByteImage::Ptr();

    int hmax = 1, vmax = 1;
    for (Component const& c : d.comps) {
        hmax = c.h > hmax ? c.h : hmax;
        vmax = c.v > vmax ? c.v : vmax;
    }
    int const mcux = (d.width + 8 * hmax - 1) / (8 * hmax);
    int const mcuy = (d.height + 8 * vmax - 1) / (8 * vmax);
    for (Component& c : d.comps) {
        if (hmax % c.h != 0 || vmax % c.v != 0)
            d.fail("fractional sampling ratios are not supported");
        c.width_blocks = mcux * c.h;
        c.height_blocks = mcuy * c.v;
        // jdmaster.c: downsampled_width = ceil(image_width * h_samp / max_h_samp)
        c.down_w = (int)(((long)d.width * c.h + hmax - 1) / hmax);
        c.down_h = (int)(((long)d.height * c.v + vmax - 1) / vmax);
        c.coefs.assign((std::size_t)c.width_blocks * c.height_blocks * 64, 0);
    }
    // every scan of the frame (one for most sequential files, about ten for a
    // progressive one), then the inverse DCT of what has arrived
    // (a progressive file has about ten scans; a file of thousands of empty
    // ones would walk every block of the frame once per scan)
    int scans = 0;
    do {
        if (++scans > 256)
            d.fail("more than 256 scans");
        decode_scan(d, sc, mcux, mcuy);
    } while (next_scan(d, &sc));
    for (Component& c : d.comps) {
        if (!c.q_latched)
            d.fail("a component of the frame is in no scan");
        std::size_t const stride = (std::size_t)c.width_blocks * 8;
        c.plane.assign(stride * c.height_blocks * 8, 0);
        for (int by = 0; by < c.height_blocks; ++by)
            for (int bx = 0; bx < c.width_blocks; ++bx) {
                const int16_t* src = c.coefs.data()
                    + ((std::size_t)by * c.width_blocks + bx) * 64;
                int coef[64];
                for (int i = 0; i < 64; ++i)
                    coef[i] = (int)src[i] * c.q[i];
                idct_islow(coef, c.plane.data() + (std::size_t)by * 8 * stride
                    + (std::size_t)bx * 8, stride);
            }
        std::vector<int16_t>().swap(c.coefs);
    }

    // colour space (jdapimin.c, default_decompress_parms)
    bool ycc = false;
    if (n == 3) {
        if (d.saw_jfif)
            ycc = true;
        else if (d.saw_adobe)
            ycc = d.adobe_transform == 1;
        else
            ycc = !(d.comps[0].id == 'R' && d.comps[1].id == 'G' && d.comps[2].id == 'B');
        if (d.saw_adobe && !d.saw_jfif && d.adobe_transform != 0 && d.adobe_transform != 1)
            ycc = true;
    }

    ByteImage::Ptr img = ByteImage::create_for_overwrite(d.width, d.height, n);
    unsigned char* dst = img->begin();
    if (n == 1) {
        Component const& c = d.comps[0];
        std::size_t const stride = (std::size_t)c.width_blocks * 8;
        for (int y = 0; y < d.height; ++y)
            std::memcpy(dst + (std::size_t)y * d.width, c.plane.data() + (std::size_t)y * stride,
                (std::size_t)d.width);
        return img;
    }
    std::vector<unsigned char> full[3];
    std::size_t fw[3];
    for (int k = 0; k < 3; ++k) {
        Component const& c = d.comps[(std::size_t)k];
        upsample(c, hmax / c.h, vmax / c.v, d.width, d.height, &full[k]);
        fw[k] = (std::size_t)c.down_w * (hmax / c.h);
    }
    // jdcolor.c: build_ycc_rgb_table, SCALEBITS 16
    int cr_r[256], cb_b[256];
    long cr_g[256], cb_g[256];
    for (int i = 0; i < 256; ++i) {
        long const x = i - 128;
        cr_r[i] = (int)((91881L * x + 32768L) >> 16);     // FIX(1.40200)
        cb_b[i] = (int)((116130L * x + 32768L) >> 16);    // FIX(1.77200)
        cr_g[i] = -46802L * x;                            // FIX(0.71414)
        cb_g[i] = -22554L * x + 32768L;                   // FIX(0.34414)
    }
    for (int y = 0; y < d.height; ++y)
        for (int x = 0; x < d.width; ++x) {
            int const a = full[0][(std::size_t)y * fw[0] + x];
            int const b = full[1][(std::size_t)y * fw[1] + x];
            int const c = full[2][(std::size_t)y * fw[2] + x];
            unsigned char* o = dst + ((std::size_t)y * d.width + x) * 3;
            if (ycc) {
                o[0] = clamp_u8(a + cr_r[c]);
                o[1] = clamp_u8(a + (int)((cb_g[b] + cr_g[c]) >> 16));
                o[2] = clamp_u8(a + cb_b[b]);
            } else {
                o[0] = (unsigned char)a;
                o[1] = (unsigned char)b;
                o[2] = (unsigned char)c;
            }
        }
    return img;
}

} // namespace

ByteImage::Ptr
load_jpeg_u8(std::string const& path)
{
    return decode(path, false, nullptr);
}

bool
jpeg_header(std::string const& path, int* whc)
{
    try {
        decode(path, true, whc);
        return true;
    } catch (std::exception const&) {
        return false;
    }
}

} // namespace smvs_amd
