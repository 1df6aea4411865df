// Visualize.cpp -- see Visualize.h.  Parity with OpenCV's rendering is unpinned (GUI path, out of the hot path).
#include "Visualize.h"
#include <zlib.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <stdexcept>

static inline float clamp01(float v) { return std::min(1.f, std::max(0.f, v)); }

RgbImage normals_image(const float* N, const std::vector<int>& imask, int rows, int cols) {
    RgbImage im; im.rows = rows; im.cols = cols; im.px.assign((size_t)rows * cols * 3, 0.f);
    const size_t P = imask.size();
    for (size_t p = 0; p < P; ++p) {
        const int r = imask[p] % rows, c = imask[p] / rows;                           // Utilities.cpp:285-286
        float* o = &im.px[((size_t)r * cols + c) * 3];
        o[0] = clamp01(0.5f + 0.5f * N[p]);                                           // Utilities.cpp:288-290 (shown as R,G,B)
        o[1] = clamp01(0.5f + 0.5f * N[P + p]);
        o[2] = clamp01(0.5f - 0.5f * N[2 * P + p]);
    }
    float mn = 1e30f, mx = -1e30f;                                                    // cv::normalize(..., 0, 1, MINMAX), Utilities.cpp:294
    for (float v : im.px) { mn = std::min(mn, v); mx = std::max(mx, v); }
    if (mx > mn) for (float& v : im.px) v = (v - mn) / (mx - mn);
    return im;
}

RgbImage albedo_image(const float* rho, const std::vector<int>& imask, int rows, int cols, int nchannels) {
    RgbImage im; im.rows = rows; im.cols = cols; im.px.assign((size_t)rows * cols * 3, 0.f);
    const size_t P = imask.size();
    std::vector<float> cap(nchannels, 1.f);
    for (int c = 0; c < nchannels; ++c) {                                             // median + 5 sigma per channel, Utilities.cpp:247-262
        std::vector<float> ch(rho + (size_t)c * P, rho + (size_t)(c + 1) * P);
        const float sum = std::accumulate(ch.begin(), ch.end(), 0.f), mean = sum / ch.size();
        const float sq = std::inner_product(ch.begin(), ch.end(), ch.begin(), 0.f);
        const float sd = std::sqrt(std::max(0.f, sq / ch.size() - mean * mean));
        std::sort(ch.begin(), ch.end());
        const float med = (P % 2 == 0) ? 0.5f * (ch[P / 2 - 1] + ch[P / 2]) : ch[P / 2];
        cap[c] = med + 5 * sd;
    }
    for (size_t p = 0; p < P; ++p) {
        const int r = imask[p] % rows, c = imask[p] / rows;
        float* o = &im.px[((size_t)r * cols + c) * 3];
        for (int k = 0; k < 3; ++k) { const int ch = std::min(k, nchannels - 1); o[k] = clamp01(std::min(cap[ch], rho[(size_t)ch * P + p])); }   // Utilities.cpp:268-272
    }
    return im;
}

// MATLAB/OpenCV "bone" colour map: (7 gray + hot with channels reversed) / 8
static void bone(float x, float* rgb) {
    auto hot_r = clamp01(x / 0.375f), hot_g = clamp01((x - 0.375f) / 0.375f), hot_b = clamp01((x - 0.75f) / 0.25f);
    rgb[0] = (7 * x + hot_b) / 8; rgb[1] = (7 * x + hot_g) / 8; rgb[2] = (7 * x + hot_r) / 8;
}

RgbImage depth_image(const float* z, const std::vector<int>& imask, int rows, int cols) {
    RgbImage im; im.rows = rows; im.cols = cols; im.px.assign((size_t)rows * cols * 3, 0.f);
    const size_t P = imask.size();
    float mn = 1e30f, mx = -1e30f;
    for (size_t p = 0; p < P; ++p) { mn = std::min(mn, -z[p]); mx = std::max(mx, -z[p]); }   // -z, min-max normalised, Utilities.cpp:304-307
    for (size_t p = 0; p < P; ++p) {
        const int r = imask[p] % rows, c = imask[p] / rows;
        const float t = mx > mn ? (-z[p] - mn) / (mx - mn) : 0.f;
        bone(std::floor(t * 255.f) / 255.f, &im.px[((size_t)r * cols + c) * 3]);              // 8-bit quantisation before the LUT, Utilities.cpp:308-310
    }
    return im;
}

RgbImage resize_bilinear(const RgbImage& s, float scale) {
    RgbImage d; d.rows = std::max(1, (int)std::lround(s.rows * scale)); d.cols = std::max(1, (int)std::lround(s.cols * scale));
    d.px.resize((size_t)d.rows * d.cols * 3);
    const float fy = (float)s.rows / d.rows, fx = (float)s.cols / d.cols;
    for (int i = 0; i < d.rows; ++i) {
        const float sy = (i + 0.5f) * fy - 0.5f; const int y0 = (int)std::floor(sy); const float wy = sy - y0;
        const int ya = std::min(std::max(y0, 0), s.rows - 1), yb = std::min(std::max(y0 + 1, 0), s.rows - 1);
        for (int j = 0; j < d.cols; ++j) {
            const float sx = (j + 0.5f) * fx - 0.5f; const int x0 = (int)std::floor(sx); const float wx = sx - x0;
            const int xa = std::min(std::max(x0, 0), s.cols - 1), xb = std::min(std::max(x0 + 1, 0), s.cols - 1);
            for (int k = 0; k < 3; ++k) {
                const float a = s.px[((size_t)ya * s.cols + xa) * 3 + k], b = s.px[((size_t)ya * s.cols + xb) * 3 + k];
                const float c = s.px[((size_t)yb * s.cols + xa) * 3 + k], e = s.px[((size_t)yb * s.cols + xb) * 3 + k];
                d.px[((size_t)i * d.cols + j) * 3 + k] = (1 - wy) * ((1 - wx) * a + wx * b) + wy * ((1 - wx) * c + wx * e);
            }
        }
    }
    return d;
}

static void put_be32(std::vector<uint8_t>& b, uint32_t v) { b.push_back(v >> 24); b.push_back(v >> 16); b.push_back(v >> 8); b.push_back(v); }
static void chunk(std::vector<uint8_t>& out, const char* type, const std::vector<uint8_t>& data) {
    put_be32(out, (uint32_t)data.size());
    const size_t start = out.size();
    out.insert(out.end(), type, type + 4); out.insert(out.end(), data.begin(), data.end());
    put_be32(out, (uint32_t)crc32(0L, out.data() + start, (uInt)(out.size() - start)));
}

void png_write_rgb8(const std::string& path, const RgbImage& im) {
    std::vector<uint8_t> raw((size_t)im.rows * (im.cols * 3 + 1));
    for (int i = 0; i < im.rows; ++i) {
        uint8_t* row = &raw[(size_t)i * (im.cols * 3 + 1)];
        row[0] = 0;                                                    // filter: none
        for (int j = 0; j < im.cols * 3; ++j) row[1 + j] = (uint8_t)std::lround(clamp01(im.px[(size_t)i * im.cols * 3 + j]) * 255.f);
    }
    uLongf n = compressBound((uLong)raw.size());
    std::vector<uint8_t> z(n);
    if (compress2(z.data(), &n, raw.data(), (uLong)raw.size(), 6) != Z_OK) throw std::runtime_error("png_write: deflate failed");
    z.resize(n);
    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    std::vector<uint8_t> ihdr; put_be32(ihdr, im.cols); put_be32(ihdr, im.rows);
    ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);      // 8 bit, RGB
    chunk(out, "IHDR", ihdr); chunk(out, "IDAT", z); chunk(out, "IEND", {});
    FILE* f = fopen(path.c_str(), "wb");
    if (!f || fwrite(out.data(), 1, out.size(), f) != out.size()) { if (f) fclose(f); throw std::runtime_error("cannot write " + path); }
    fclose(f);
}
