// PngIO.cpp -- minimal PNG decoder: non-interlaced, colour types 0/2/3/4/6, bit depths 1-16.
#include "PngIO.h"
#include <zlib.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <stdexcept>

namespace {
uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
int paeth(int a, int b, int c) {
    const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}
}  // namespace

PngImage png_read(const std::string& path) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("ERROR: Could not load image " + path);              // Utilities.cpp:325-327
    std::vector<uint8_t> buf;
    uint8_t tmp[65536];
    size_t n;
    while ((n = fread(tmp, 1, sizeof(tmp), f)) > 0) buf.insert(buf.end(), tmp, tmp + n);
    fclose(f);
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (buf.size() < 8 || memcmp(buf.data(), sig, 8) != 0) throw std::runtime_error(path + ": not a PNG file");
    int w = 0, h = 0, depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, plte;
    size_t off = 8;
    while (off + 12 <= buf.size()) {
        const uint32_t len = be32(&buf[off]);
        const char* type = (const char*)&buf[off + 4];
        const uint8_t* d = &buf[off + 8];
        if (off + 12 + len > buf.size()) throw std::runtime_error(path + ": truncated PNG chunk");
        if (!memcmp(type, "IHDR", 4)) { w = (int)be32(d); h = (int)be32(d + 4); depth = d[8]; ctype = d[9]; interlace = d[12]; }
        else if (!memcmp(type, "PLTE", 4)) plte.assign(d, d + len);
        else if (!memcmp(type, "IDAT", 4)) idat.insert(idat.end(), d, d + len);
        else if (!memcmp(type, "IEND", 4)) break;
        off += 12 + len;
    }
    if (w <= 0 || h <= 0) throw std::runtime_error(path + ": missing IHDR");
    if (interlace) throw std::runtime_error(path + ": interlaced PNGs are not supported");
    int ch;
    switch (ctype) { case 0: ch = 1; break; case 2: ch = 3; break; case 3: ch = 1; break; case 4: ch = 2; break; case 6: ch = 4; break;
        default: throw std::runtime_error(path + ": bad PNG colour type"); }
    const size_t bpp_bits = (size_t)ch * depth;
    const size_t stride = (w * bpp_bits + 7) / 8;
    const size_t bpp = std::max<size_t>(1, bpp_bits / 8);
    std::vector<uint8_t> raw((stride + 1) * (size_t)h);
    uLongf dl = (uLongf)raw.size();
    if (uncompress(raw.data(), &dl, idat.data(), (uLong)idat.size()) != Z_OK || dl != raw.size())
        throw std::runtime_error(path + ": corrupt PNG image data");
    std::vector<uint8_t> img(stride * (size_t)h);
    for (int y = 0; y < h; ++y) {                                  // undo the per-scanline filters
        const uint8_t ft = raw[(stride + 1) * y];
        const uint8_t* in = &raw[(stride + 1) * y + 1];
        uint8_t* out = &img[stride * y];
        const uint8_t* up = y ? &img[stride * (y - 1)] : nullptr;
        for (size_t x = 0; x < stride; ++x) {
            const int a = x >= bpp ? out[x - bpp] : 0, b = up ? up[x] : 0, c = (up && x >= bpp) ? up[x - bpp] : 0;
            int v = in[x];
            switch (ft) { case 0: break; case 1: v += a; break; case 2: v += b; break; case 3: v += (a + b) / 2; break; case 4: v += paeth(a, b, c); break;
                default: throw std::runtime_error(path + ": bad PNG filter"); }
            out[x] = (uint8_t)v;
        }
    }
    PngImage im;
    im.width = w; im.height = h;
    im.channels = (ctype == 3) ? 3 : ch;
    im.bit_depth = depth == 16 ? 16 : 8;
    im.pix.resize((size_t)w * h * im.channels);
    for (int y = 0; y < h; ++y) {
        const uint8_t* row = &img[stride * y];
        for (int x = 0; x < w; ++x)
            for (int k = 0; k < ch; ++k) {
                const size_t s = (size_t)x * ch + k;
                unsigned v;
                if (depth == 16) v = ((unsigned)row[2 * s] << 8) | row[2 * s + 1];
                else if (depth == 8) v = row[s];
                else { const size_t bit = s * depth; v = (row[bit / 8] >> (8 - depth - bit % 8)) & ((1u << depth) - 1); if (ctype != 3) v = v * 255u / ((1u << depth) - 1); }
                if (ctype == 3) {
                    if (3 * v + 2 >= plte.size()) throw std::runtime_error(path + ": palette index out of range");
                    for (int q = 0; q < 3; ++q) im.pix[((size_t)y * w + x) * 3 + q] = plte[3 * v + q];
                } else {
                    im.pix[((size_t)y * w + x) * ch + k] = (uint16_t)v;
                }
            }
    }
    return im;
}

static inline unsigned to8(const PngImage& im, unsigned v) { return im.bit_depth == 16 ? (v >> 8) : v; }

std::vector<uint8_t> png_as_rgb8(const PngImage& im) {
    std::vector<uint8_t> out((size_t)im.width * im.height * 3);
    for (size_t p = 0; p < (size_t)im.width * im.height; ++p) {
        const uint16_t* s = &im.pix[p * im.channels];
        if (im.channels >= 3) for (int q = 0; q < 3; ++q) out[3 * p + q] = (uint8_t)to8(im, s[q]);
        else for (int q = 0; q < 3; ++q) out[3 * p + q] = (uint8_t)to8(im, s[0]);
    }
    return out;
}

std::vector<uint16_t> png_as_gray_native(const PngImage& im) {
    std::vector<uint16_t> out((size_t)im.width * im.height);
    for (size_t p = 0; p < out.size(); ++p) {
        const uint16_t* s = &im.pix[p * im.channels];
        if (im.channels >= 3) out[p] = (uint16_t)std::lround(0.299 * s[0] + 0.587 * s[1] + 0.114 * s[2]);
        else out[p] = s[0];
    }
    return out;
}

std::vector<uint8_t> png_as_gray8(const PngImage& im) {
    std::vector<uint16_t> g = png_as_gray_native(im);
    std::vector<uint8_t> out(g.size());
    for (size_t p = 0; p < g.size(); ++p) out[p] = (uint8_t)to8(im, g[p]);
    return out;
}
