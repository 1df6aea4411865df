// Preprocess.cpp -- see Preprocess.h.  PARITY UNPINNED against OpenCV 3.3.0 (binary not available).
#include "Preprocess.h"
#include <algorithm>
#include <cmath>
#include <functional>
#include <queue>

void mean_across_channels_cpu(const float* z0, int n_pix, int nc, std::vector<float>& mean, std::vector<uint8_t>& flag) {
    mean.assign(n_pix, 0.f); flag.assign(n_pix, 0);
    for (int t = 0; t < n_pix; ++t) {
        float avg = 0.f;
        for (int c = 0; c < nc; ++c) {
            const float v = z0[(size_t)c * n_pix + t];
            if (v != 0.f) avg += v; else flag[t] = 1;                 // dc.cu:101-106
        }
        mean[t] = avg / (float)nc;                                    // dc.cu:108: divides by nc, not by the valid count
    }
}

// ---- Telea, "An image inpainting technique based on the fast marching method" (2004) ----------
namespace {
enum : uint8_t { KNOWN = 0, BAND = 1, INSIDE = 2 };

float solve_eikonal(const std::vector<float>& T, const std::vector<uint8_t>& st, int rows, int cols, int i1, int j1, int i2, int j2) {
    auto ok = [&](int i, int j) { return i >= 0 && i < rows && j >= 0 && j < cols; };
    float sol = 1e6f;
    const bool a = ok(i1, j1) && st[i1 * cols + j1] == KNOWN, b = ok(i2, j2) && st[i2 * cols + j2] == KNOWN;
    if (a && b) {
        const float t1 = T[i1 * cols + j1], t2 = T[i2 * cols + j2];
        const float r = std::sqrt(std::max(0.f, 2.f - (t1 - t2) * (t1 - t2)));
        float s = (t1 + t2 - r) * 0.5f;
        if (s >= t1 && s >= t2) sol = s;
        else { s += r; if (s >= t1 && s >= t2) sol = s; }
    } else if (a) sol = 1.f + T[i1 * cols + j1];
    else if (b) sol = 1.f + T[i2 * cols + j2];
    return sol;
}
}  // namespace

void inpaint_telea(std::vector<float>& img, const std::vector<uint8_t>& flag, int rows, int cols, int radius) {
    const int n = rows * cols;
    bool any = false;
    for (int t = 0; t < n; ++t) any |= flag[t] != 0;
    if (!any) return;
    std::vector<uint8_t> st(n, KNOWN);
    std::vector<float> T(n, 0.f);
    typedef std::pair<float, int> Node;
    std::priority_queue<Node, std::vector<Node>, std::greater<Node>> heap;
    for (int t = 0; t < n; ++t) if (flag[t]) { st[t] = INSIDE; T[t] = 1e6f; }
    const int di[4] = {-1, 1, 0, 0}, dj[4] = {0, 0, -1, 1};
    for (int i = 0; i < rows; ++i)                       // narrow band: known pixels that touch the hole
        for (int j = 0; j < cols; ++j) {
            if (st[i * cols + j] != KNOWN) continue;
            for (int k = 0; k < 4; ++k) {
                const int a = i + di[k], b = j + dj[k];
                if (a >= 0 && a < rows && b >= 0 && b < cols && flag[a * cols + b]) { st[i * cols + j] = BAND; heap.push(Node(0.f, i * cols + j)); break; }
            }
        }
    // distance to the boundary on the KNOWN side too (needed by the level-set weight), limited to `radius`
    std::vector<float> Tout(n, 1e6f);
    {
        std::priority_queue<Node, std::vector<Node>, std::greater<Node>> h2;
        std::vector<uint8_t> s2(n, INSIDE);
        for (int t = 0; t < n; ++t) {
            if (st[t] == BAND) { s2[t] = BAND; Tout[t] = 0.f; h2.push(Node(0.f, t)); }
            else if (st[t] == INSIDE) { s2[t] = KNOWN; Tout[t] = 0.f; }
        }
        while (!h2.empty()) {
            const Node nd = h2.top(); h2.pop();
            const int t = nd.second;
            if (s2[t] == KNOWN) continue;
            s2[t] = KNOWN;
            if (nd.first > (float)radius) continue;
            const int i = t / cols, j = t % cols;
            for (int k = 0; k < 4; ++k) {
                const int a = i + di[k], b = j + dj[k];
                if (a < 0 || a >= rows || b < 0 || b >= cols || s2[a * cols + b] == KNOWN) continue;
                const float v = std::min(std::min(solve_eikonal(Tout, s2, rows, cols, a - 1, b, a, b - 1), solve_eikonal(Tout, s2, rows, cols, a + 1, b, a, b - 1)),
                                         std::min(solve_eikonal(Tout, s2, rows, cols, a - 1, b, a, b + 1), solve_eikonal(Tout, s2, rows, cols, a + 1, b, a, b + 1)));
                if (v < Tout[a * cols + b]) { Tout[a * cols + b] = v; s2[a * cols + b] = BAND; h2.push(Node(v, a * cols + b)); }
            }
        }
        for (int t = 0; t < n; ++t) if (st[t] != INSIDE) T[t] = -Tout[t];   // negative outside, as in the paper
        for (int t = 0; t < n; ++t) if (st[t] == BAND) T[t] = 0.f;
    }
    auto Tat = [&](int i, int j) { i = std::min(std::max(i, 0), rows - 1); j = std::min(std::max(j, 0), cols - 1); return T[i * cols + j]; };
    while (!heap.empty()) {
        const Node nd = heap.top(); heap.pop();
        const int t = nd.second;
        if (st[t] == KNOWN) continue;
        st[t] = KNOWN;
        const int i = t / cols, j = t % cols;
        for (int k = 0; k < 4; ++k) {
            const int a = i + di[k], b = j + dj[k];
            if (a < 0 || a >= rows || b < 0 || b >= cols) continue;
            const int q = a * cols + b;
            if (st[q] != INSIDE) continue;
            const float v = std::min(std::min(solve_eikonal(T, st, rows, cols, a - 1, b, a, b - 1), solve_eikonal(T, st, rows, cols, a + 1, b, a, b - 1)),
                                     std::min(solve_eikonal(T, st, rows, cols, a - 1, b, a, b + 1), solve_eikonal(T, st, rows, cols, a + 1, b, a, b + 1)));
            T[q] = v;
            // inpaint q from the known pixels within `radius` (eq. 2-4 of the paper)
            const float gTx = 0.5f * (Tat(a, b + 1) - Tat(a, b - 1)), gTy = 0.5f * (Tat(a + 1, b) - Tat(a - 1, b));
            double Ia = 0.0, sw = 0.0;
            for (int u = a - radius; u <= a + radius; ++u)
                for (int w2 = b - radius; w2 <= b + radius; ++w2) {
                    if (u < 0 || u >= rows || w2 < 0 || w2 >= cols) continue;
                    const int p = u * cols + w2;
                    if (st[p] == INSIDE || (u == a && w2 == b)) continue;
                    const float ry = (float)(a - u), rx = (float)(b - w2);
                    const float d2 = rx * rx + ry * ry;
                    if (d2 > (float)(radius * radius)) continue;
                    float dir = std::fabs(rx * gTx + ry * gTy) / std::sqrt(d2);
                    if (dir < 1e-6f) dir = 1e-6f;
                    const float dst = 1.f / d2;
                    const float lev = 1.f / (1.f + std::fabs(T[p] - v));
                    const float wgt = dir * dst * lev;
                    // first-order term: gradient of the image at the known pixel (central where possible)
                    float gx = 0.f, gy = 0.f;
                    if (w2 + 1 < cols && w2 - 1 >= 0 && st[p + 1] != INSIDE && st[p - 1] != INSIDE) gx = 0.5f * (img[p + 1] - img[p - 1]);
                    if (u + 1 < rows && u - 1 >= 0 && st[p + cols] != INSIDE && st[p - cols] != INSIDE) gy = 0.5f * (img[p + cols] - img[p - cols]);
                    Ia += wgt * (img[p] + gx * rx + gy * ry);
                    sw += wgt;
                }
            if (sw > 0) img[q] = (float)(Ia / sw);
            st[q] = BAND;
            heap.push(Node(v, q));
        }
    }
}

// cv::bilateralFilter(src, dst, d = -1, sigmaColor, sigmaSpace) on CV_32F: radius = round(1.5 sigmaSpace),
// circular support, BORDER_REFLECT_101; exact exponentials (OpenCV interpolates a LUT for float images)
void bilateral_filter(const std::vector<float>& src, std::vector<float>& dst, int rows, int cols, float sigma_color, float sigma_space) {
    const int radius = std::max(1, (int)std::lround(sigma_space * 1.5f));
    const float gc = -0.5f / (sigma_color * sigma_color), gs = -0.5f / (sigma_space * sigma_space);
    auto refl = [](int x, int n) { if (n == 1) return 0; while (x < 0 || x >= n) x = x < 0 ? -x : 2 * n - 2 - x; return x; };
    dst.resize(src.size());
    for (int i = 0; i < rows; ++i)
        for (int j = 0; j < cols; ++j) {
            const float v0 = src[(size_t)i * cols + j];
            float sum = 0.f, wsum = 0.f;
            for (int di = -radius; di <= radius; ++di)
                for (int dj = -radius; dj <= radius; ++dj) {
                    const float r2 = (float)(di * di + dj * dj);
                    if (r2 > (float)(radius * radius)) continue;
                    const float v = src[(size_t)refl(i + di, rows) * cols + refl(j + dj, cols)];
                    const float w = std::exp(r2 * gs + (v - v0) * (v - v0) * gc);
                    sum += w * v; wsum += w;
                }
            dst[(size_t)i * cols + j] = sum / wsum;
        }
}

// cv::resize(..., INTER_CUBIC): Keys kernel a = -0.75, src coordinate (dst + 0.5) * scale - 0.5, replicated border
void resize_cubic(const std::vector<float>& src, int rows, int cols, std::vector<float>& dst, int out_rows, int out_cols) {
    auto coeffs = [](float x, float* c) {
        const float A = -0.75f;
        c[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
        c[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
        c[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
        c[3] = 1.f - c[0] - c[1] - c[2];
    };
    const double sy = (double)rows / out_rows, sx = (double)cols / out_cols;
    std::vector<int> xi(out_cols); std::vector<float> xc((size_t)out_cols * 4);
    for (int j = 0; j < out_cols; ++j) {
        const float fx = (float)((j + 0.5) * sx - 0.5);
        const int x0 = (int)std::floor(fx);
        xi[j] = x0; coeffs(fx - x0, &xc[(size_t)j * 4]);
    }
    dst.resize((size_t)out_rows * out_cols);
    auto clampi = [](int v, int n) { return std::min(std::max(v, 0), n - 1); };
    for (int i = 0; i < out_rows; ++i) {
        const float fy = (float)((i + 0.5) * sy - 0.5);
        const int y0 = (int)std::floor(fy);
        float yc[4]; coeffs(fy - y0, yc);
        for (int j = 0; j < out_cols; ++j) {
            float acc = 0.f;
            for (int a = 0; a < 4; ++a) {
                const float* row = &src[(size_t)clampi(y0 - 1 + a, rows) * cols];
                float r = 0.f;
                for (int b = 0; b < 4; ++b) r += xc[(size_t)j * 4 + b] * row[clampi(xi[j] - 1 + b, cols)];
                acc += yc[a] * r;
            }
            dst[(size_t)i * out_cols + j] = acc;
        }
    }
}

void preprocess_depth(const float* z0, int z0_h, int z0_w, int z0_n, int I_h, int I_w, std::vector<float>& zs, std::vector<float>& z_full) {
    const int n = z0_h * z0_w;
    std::vector<uint8_t> flag;
    mean_across_channels_cpu(z0, n, z0_n, zs, flag);              // SRPS.cu:124 (GPU kernel in the reference; same arithmetic)
    // the column-major (z0_h x z0_w) buffer viewed as a row-major (z0_w x z0_h) image: SRPS.cu:130-132
    const int rows = z0_w, cols = z0_h;
    inpaint_telea(zs, flag, rows, cols, 16);                      // SRPS.cu:133
    float mx = 0.f;
    for (float v : zs) mx = std::max(mx, v);                      // SRPS.cu:137
    if (mx <= 0.f) mx = 1.f;
    for (float& v : zs) v /= mx;                                  // SRPS.cu:138
    std::vector<float> sm;
    bilateral_filter(zs, sm, rows, cols, 2.f, 2.f);               // SRPS.cu:139
    for (float& v : sm) v *= mx;                                  // SRPS.cu:140
    zs = sm;
    resize_cubic(zs, rows, cols, z_full, I_w, I_h);               // SRPS.cu:148-149: Size(I_h, I_w) = (width, height)
}
