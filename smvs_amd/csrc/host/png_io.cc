// PNG embeddings of an MVE view: mve::View stores byte
// images as <name>.png (makescene's "undistorted.png", the "undist-L<n>"
// inputs smvsrecon writes, app/smvsrecon.cc:634-647).  The reference goes
// through MVE's libpng wrapper; this tree has no libpng headers, so the
// container is read and written here on top of zlib: signature, IHDR, PLTE,
// IDAT (inflate), the five scanline filters, IEND.  8 bits per sample, colour
// types 0 (grey), 2 (RGB), 3 (palette -> RGB), 4 (grey + alpha), 6 (RGBA),
// non-interlaced and Adam7.  16-bit samples are refused (mve::ByteImage has
// none).  The writer emits filter 0 scanlines, one IDAT.
#include "png_io.h"

#include <zlib.h>

#include <cstdio>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <vector>

namespace smvs_amd {

namespace {

unsigned char const PNG_SIGNATURE[8] = { 0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A };

uint32_t
be32(unsigned char const* p)
{
    return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
}

void
put_be32(unsigned char* p, uint32_t v)
{
    p[0] = (unsigned char)(v >> 24);
    p[1] = (unsigned char)(v >> 16);
    p[2] = (unsigned char)(v >> 8);
    p[3] = (unsigned char)v;
}

struct Header
{
    int width = 0, height = 0, bit_depth = 0, colour_type = 0, interlace = 0;
    int samples(void) const
    {
        switch (colour_type) {
        case 0: return 1;
        case 2: return 3;
        case 3: return 1;
        case 4: return 2;
        case 6: return 4;
        default: return 0;
        }
    }
    // channels of the decoded image (the palette expands to RGB)
    int channels(void) const { return colour_type == 3 ? 3 : samples(); }
};

std::vector<unsigned char>
read_file(std::string const& path)
{
    std::ifstream in(path.c_str(), std::ios::binary);
    if (!in)
        throw std::runtime_error("cannot open " + path);
    in.seekg(0, std::ios::end);
    std::streamoff const size = in.tellg();
    in.seekg(0, std::ios::beg);
    std::vector<unsigned char> data((std::size_t)size);
    in.read(reinterpret_cast<char*>(data.data()), size);
    if (!in)
        throw std::runtime_error("cannot read " + path);
    return data;
}

Header
parse_ihdr(unsigned char const* d, std::string const& path)
{
    Header h;
    h.width = (int)be32(d);
    h.height = (int)be32(d + 4);
    h.bit_depth = d[8];
    h.colour_type = d[9];
    h.interlace = d[12];
    if (h.width <= 0 || h.height <= 0 || d[10] != 0 || d[11] != 0 || h.interlace > 1
        || h.samples() == 0)
        throw std::runtime_error("bad PNG header: " + path);
    // The header is untrusted input: bound the dimensions before anything is
    // allocated or any pass size is computed from them (2^20 pixels a side and
    // 2^30 samples -- far beyond any camera image -- keep every product below
    // in range and a forged header from asking for gigabytes).
    constexpr int MAX_SIDE = 1 << 20;
    constexpr long long MAX_SAMPLES = 1ll << 30;
    if (h.width > MAX_SIDE || h.height > MAX_SIDE
        || (long long)h.width * h.height * h.samples() > MAX_SAMPLES)
        throw std::runtime_error("PNG dimensions out of range: " + path);
    return h;
}

// One pass of the scanline filters (PNG specification, section 9), in place:
// `rows` scanlines of `stride` bytes, each preceded by its filter byte.
void
unfilter(unsigned char* data, int rows, std::size_t stride, int bpp,
    std::string const& path)
{
    std::vector<unsigned char> zero(stride, 0);
    unsigned char const* prev = zero.data();
    for (int y = 0; y < rows; ++y) {
        unsigned char* line = data + (std::size_t)y * (stride + 1);
        int const filter = line[0];
        unsigned char* cur = line + 1;
        for (std::size_t i = 0; i < stride; ++i) {
            int const a = i >= (std::size_t)bpp ? cur[i - bpp] : 0;
            int const b = prev[i];
            int const c = i >= (std::size_t)bpp ? prev[i - bpp] : 0;
            int add = 0;
            switch (filter) {
            case 0: break;
            case 1: add = a; break;
            case 2: add = b; break;
            case 3: add = (a + b) >> 1; break;
            case 4: {
                int const p = a + b - c;
                int const pa = p > a ? p - a : a - p;
                int const pb = p > b ? p - b : b - p;
                int const pc = p > c ? p - c : c - p;
                add = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
                break;
            }
            default:
                throw std::runtime_error("bad PNG filter type: " + path);
            }
            cur[i] = (unsigned char)(cur[i] + add);
        }
        prev = cur;
    }
}

} // namespace

bool
png_header(std::string const& path, int* whc)
{
    std::ifstream in(path.c_str(), std::ios::binary);
    if (!in)
        return false;
    unsigned char head[33];
    in.read(reinterpret_cast<char*>(head), sizeof(head));
    if (!in || std::memcmp(head, PNG_SIGNATURE, 8) != 0 || be32(head + 8) != 13
        || std::memcmp(head + 12, "IHDR", 4) != 0)
        return false;
    try {
        Header const h = parse_ihdr(head + 16, path);
        whc[0] = h.width;
        whc[1] = h.height;
        whc[2] = h.channels();
    } catch (std::exception const&) {
        return false;
    }
    return true;
}

ByteImage::Ptr
load_png_u8(std::string const& path)
{
    std::vector<unsigned char> const file = read_file(path);
    if (file.size() < 8 || std::memcmp(file.data(), PNG_SIGNATURE, 8) != 0)
        throw std::runtime_error("not a PNG file: " + path);
    Header h;
    bool have_header = false, ended = false;
    std::vector<unsigned char> idat, palette;
    std::size_t pos = 8;
    while (!ended) {
        if (pos + 12 > file.size())
            throw std::runtime_error("truncated PNG file: " + path);
        std::size_t const len = be32(&file[pos]);
        unsigned char const* type = &file[pos + 4];
        if (pos + 12 + len > file.size())
            throw std::runtime_error("truncated PNG chunk: " + path);
        unsigned char const* body = &file[pos + 8];
        uint32_t const crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), type, (uInt)(len + 4));
        if (crc != be32(body + len))
            throw std::runtime_error("PNG chunk checksum mismatch: " + path);
        if (std::memcmp(type, "IHDR", 4) == 0) {
            if (len != 13)
                throw std::runtime_error("bad PNG header: " + path);
            h = parse_ihdr(body, path);
            have_header = true;
        } else if (std::memcmp(type, "PLTE", 4) == 0) {
            palette.assign(body, body + len);
        } else if (std::memcmp(type, "IDAT", 4) == 0) {
            idat.insert(idat.end(), body, body + len);
        } else if (std::memcmp(type, "IEND", 4) == 0) {
            ended = true;
        }
        pos += 12 + len;
    }
    if (!have_header || idat.empty())
        throw std::runtime_error("PNG without header or data: " + path);
    if (h.bit_depth != 8)
        throw std::runtime_error("PNG with " + std::to_string(h.bit_depth)
            + " bits per sample (only 8-bit images are byte images): " + path);
    if (h.colour_type == 3 && palette.size() < 3)
        throw std::runtime_error("palette PNG without a palette: " + path);

    int const bpp = h.samples();
    // the passes: one for a plain image, Adam7's seven otherwise
    struct Pass { int x0, y0, dx, dy; };
    static Pass const adam7[7] = { { 0, 0, 8, 8 }, { 4, 0, 8, 8 }, { 0, 4, 4, 8 },
        { 2, 0, 4, 4 }, { 0, 2, 2, 4 }, { 1, 0, 2, 2 }, { 0, 1, 1, 2 } };
    static Pass const whole[1] = { { 0, 0, 1, 1 } };
    Pass const* passes = h.interlace ? adam7 : whole;
    int const num_passes = h.interlace ? 7 : 1;
    std::size_t raw_size = 0;
    for (int p = 0; p < num_passes; ++p) {
        int const pw = (h.width - passes[p].x0 + passes[p].dx - 1) / passes[p].dx;
        int const ph = (h.height - passes[p].y0 + passes[p].dy - 1) / passes[p].dy;
        if (pw > 0 && ph > 0)
            raw_size += (std::size_t)ph * ((std::size_t)pw * bpp + 1);
    }
    std::vector<unsigned char> raw(raw_size);
    uLongf out_len = (uLongf)raw_size;
    int const zrc = uncompress(raw.data(), &out_len, idat.data(), (uLong)idat.size());
    if (zrc != Z_OK || out_len != raw_size)
        throw std::runtime_error("PNG data does not inflate to the image size: " + path);

    int const channels = h.channels();
    ByteImage::Ptr img = ByteImage::create_for_overwrite(h.width, h.height, channels);
    uint8_t* dst = img->begin();
    std::size_t offset = 0;
    for (int p = 0; p < num_passes; ++p) {
        Pass const& ps = passes[p];
        int const pw = (h.width - ps.x0 + ps.dx - 1) / ps.dx;
        int const ph = (h.height - ps.y0 + ps.dy - 1) / ps.dy;
        if (pw <= 0 || ph <= 0)
            continue;
        std::size_t const stride = (std::size_t)pw * bpp;
        unfilter(raw.data() + offset, ph, stride, bpp, path);
        for (int y = 0; y < ph; ++y) {
            unsigned char const* line = raw.data() + offset + (std::size_t)y * (stride + 1) + 1;
            int const oy = ps.y0 + y * ps.dy;
            for (int x = 0; x < pw; ++x) {
                int const ox = ps.x0 + x * ps.dx;
                uint8_t* out = dst + ((std::size_t)oy * h.width + ox) * channels;
                if (h.colour_type == 3) {
                    std::size_t const e = (std::size_t)line[x] * 3;
                    if (e + 2 >= palette.size())
                        throw std::runtime_error("PNG palette index out of range: " + path);
                    out[0] = palette[e];
                    out[1] = palette[e + 1];
                    out[2] = palette[e + 2];
                } else {
                    for (int c = 0; c < bpp; ++c)
                        out[c] = line[(std::size_t)x * bpp + c];
                }
            }
        }
        offset += (std::size_t)ph * (stride + 1);
    }
    return img;
}

void
save_png_u8(std::string const& path, ByteImage::ConstPtr image)
{
    int const w = image->width(), h = image->height(), c = image->channels();
    int colour_type;
    switch (c) {
    case 1: colour_type = 0; break;
    case 2: colour_type = 4; break;
    case 3: colour_type = 2; break;
    case 4: colour_type = 6; break;
    default:
        throw std::invalid_argument("save_png_u8: 1 to 4 channels");
    }
    std::size_t const stride = (std::size_t)w * c;
    std::vector<unsigned char> raw((std::size_t)h * (stride + 1));
    for (int y = 0; y < h; ++y) {
        raw[(std::size_t)y * (stride + 1)] = 0;   // filter type None
        std::memcpy(&raw[(std::size_t)y * (stride + 1) + 1],
            image->begin() + (std::size_t)y * stride, stride);
    }
    uLongf zlen = compressBound((uLong)raw.size());
    std::vector<unsigned char> z(zlen);
    if (compress2(z.data(), &zlen, raw.data(), (uLong)raw.size(), 6) != Z_OK)
        throw std::runtime_error("save_png_u8: deflate failed");
    z.resize(zlen);

    std::ofstream out(path.c_str(), std::ios::binary);
    if (!out)
        throw std::runtime_error("cannot write " + path);
    auto chunk = [&](char const* type, unsigned char const* data, std::size_t len) {
        std::vector<unsigned char> buf(len + 12);
        put_be32(buf.data(), (uint32_t)len);
        std::memcpy(buf.data() + 4, type, 4);
        if (len > 0)
            std::memcpy(buf.data() + 8, data, len);
        put_be32(buf.data() + 8 + len,
            (uint32_t)crc32(crc32(0L, Z_NULL, 0), buf.data() + 4, (uInt)(len + 4)));
        out.write(reinterpret_cast<char const*>(buf.data()), (std::streamsize)buf.size());
    };
    out.write(reinterpret_cast<char const*>(PNG_SIGNATURE), 8);
    unsigned char ihdr[13];
    put_be32(ihdr, (uint32_t)w);
    put_be32(ihdr + 4, (uint32_t)h);
    ihdr[8] = 8;
    ihdr[9] = (unsigned char)colour_type;
    ihdr[10] = ihdr[11] = ihdr[12] = 0;
    chunk("IHDR", ihdr, 13);
    chunk("IDAT", z.data(), z.size());
    chunk("IEND", nullptr, 0);
    if (!out)
        throw std::runtime_error("write failed: " + path);
}

} // namespace smvs_amd
