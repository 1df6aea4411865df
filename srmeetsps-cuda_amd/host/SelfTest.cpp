// SelfTest.cpp -- the CPU pieces of the host (MAT5 and PNG readers / writers, depth pre-processing, views) driven once with
// small and with malformed inputs.  Built with -fsanitize=address,undefined (make -C srmeetsps-cuda_amd/host sanitize) and run
// by tests/test_host_sanitizers.py: the sanitizers that cannot run on the GPU pool run here, on the code that parses files.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>
#include "MatIO.h"
#include "PngIO.h"
#include "Preprocess.h"
#include "Visualize.h"

#define REQUIRE(c) do { if (!(c)) { fprintf(stderr, "SelfTest: %s failed (line %d)\n", #c, __LINE__); return 1; } } while (0)

template <typename F>
static bool throws(F f) {
    try { f(); } catch (const std::exception&) { return true; }
    return false;
}

int main(int argc, char** argv) {
    const std::string dir = argc > 1 ? argv[1] : "/tmp";
    // MAT5 round trip, then truncated and corrupted files: the reader must throw, never read out of bounds
    std::vector<float> v(1000);
    for (size_t i = 0; i < v.size(); ++i) v[i] = std::sin(0.1f * (float)i);
    const std::string mat = dir + "/selftest.mat";
    mat5_write_single(mat.c_str(), "x", v.data(), v.size());
    auto vars = mat5_read(mat);
    REQUIRE(vars.count("x") == 1 && vars["x"].data.size() == v.size() && vars["x"].data[17] == v[17]);
    std::string bytes;
    { std::ifstream f(mat, std::ios::binary); bytes.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>()); }
    for (size_t cut : {(size_t)0, (size_t)64, (size_t)130, bytes.size() / 2, bytes.size() - 3}) {
        const std::string p = dir + "/selftest_cut.mat";
        { std::ofstream f(p, std::ios::binary); f.write(bytes.data(), (std::streamsize)cut); }
        (void)throws([&] { (void)mat5_read(p); });           // either outcome is fine as long as nothing is read out of bounds
    }
    for (size_t pos : {(size_t)128, (size_t)132, (size_t)136, (size_t)152, (size_t)168}) {
        if (pos + 4 > bytes.size()) continue;
        std::string b = bytes;
        b[pos] = (char)0xff; b[pos + 1] = (char)0xff; b[pos + 2] = (char)0xff; b[pos + 3] = (char)0x7f;      // absurd sizes / types
        const std::string p = dir + "/selftest_bad.mat";
        { std::ofstream f(p, std::ios::binary); f.write(b.data(), (std::streamsize)b.size()); }
        (void)throws([&] { (void)mat5_read(p); });
    }
    // views -> PNG -> reader, then a truncated PNG
    const int rows = 24, cols = 20;
    std::vector<int> imask;
    for (int i = 0; i < rows * cols; ++i) if ((i * 7) % 5 != 0) imask.push_back(i);
    const size_t P = imask.size();
    std::vector<float> N(4 * P), rho(3 * P), z(P);
    for (size_t p = 0; p < P; ++p) {
        N[p] = 0.3f; N[P + p] = -0.2f; N[2 * P + p] = -0.93f; N[3 * P + p] = 1.f;
        rho[p] = 0.2f + 0.001f * (float)p; rho[P + p] = 0.5f; rho[2 * P + p] = 0.9f; z[p] = 1.f + 0.01f * (float)(p % 17);
    }
    const std::string png = dir + "/selftest.png";
    png_write_rgb8(png, resize_bilinear(normals_image(N.data(), imask, rows, cols), 0.5f));
    png_write_rgb8(png, albedo_image(rho.data(), imask, rows, cols, 3));
    png_write_rgb8(png, depth_image(z.data(), imask, rows, cols));
    PngImage im = png_read(png);
    REQUIRE(im.width == cols && im.height == rows && im.channels == 3);
    REQUIRE(png_as_rgb8(im).size() == (size_t)rows * cols * 3 && png_as_gray8(im).size() == (size_t)rows * cols);
    { std::ifstream f(png, std::ios::binary); bytes.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>()); }
    for (size_t cut : {(size_t)0, (size_t)8, (size_t)20, (size_t)40, bytes.size() / 2, bytes.size() - 5}) {
        const std::string p = dir + "/selftest_cut.png";
        { std::ofstream f(p, std::ios::binary); f.write(bytes.data(), (std::streamsize)cut); }
        (void)throws([&] { (void)png_read(p); });
    }
    // depth pre-processing with invalid samples (zeros): mean, inpainting, bilateral, cubic resize
    const int zh = 16, zw = 12, zn = 3, sf = 2;
    std::vector<float> z0((size_t)zn * zh * zw);
    for (int k = 0; k < zn; ++k)
        for (int t = 0; t < zh * zw; ++t) z0[(size_t)k * zh * zw + t] = (t % 11 == 0) ? 0.f : 700.f + (float)(t % 13) + (float)k;
    std::vector<float> zs, zf;
    preprocess_depth(z0.data(), zh, zw, zn, zh * sf, zw * sf, zs, zf);
    REQUIRE(zs.size() == (size_t)zh * zw && zf.size() == (size_t)zh * sf * zw * sf);
    for (float t : zf) REQUIRE(std::isfinite(t) && t > 600.f && t < 800.f);
    printf("SelfTest ok\n");
    return 0;
}
