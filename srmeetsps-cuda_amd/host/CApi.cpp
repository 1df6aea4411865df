// CApi.cpp -- extern "C" view of the host-side (CPU) pieces so that the Python tests can call exactly
// the C++ the command-line program runs: the loaders and the depth pre-processing.
#include <cstring>
#include <string>
#include "MatIO.h"
#include "PngIO.h"
#include "Preprocess.h"
#include "Utilities.h"
#include "Visualize.h"

static thread_local std::string g_host_err;
#define HOST_TRY(...) try { __VA_ARGS__; return 0; } catch (const std::exception& e) { g_host_err = e.what(); return 1; }

extern "C" {
const char* srps_host_last_error(void) { return g_host_err.c_str(); }

// zs [z0_h*z0_w], z_full [I_h*I_w], both column-major like the inputs
int srps_host_preprocess_depth(const float* z0, int z0_h, int z0_w, int z0_n, int I_h, int I_w, float* zs, float* z_full) {
    HOST_TRY({
        std::vector<float> a, b;
        preprocess_depth(z0, z0_h, z0_w, z0_n, I_h, I_w, a, b);
        memcpy(zs, a.data(), a.size() * sizeof(float));
        memcpy(z_full, b.data(), b.size() * sizeof(float));
    })
}
int srps_host_inpaint(float* img, const unsigned char* flag, int rows, int cols, int radius) {
    HOST_TRY({
        std::vector<float> a(img, img + (size_t)rows * cols);
        std::vector<uint8_t> f(flag, flag + (size_t)rows * cols);
        inpaint_telea(a, f, rows, cols, radius);
        memcpy(img, a.data(), a.size() * sizeof(float));
    })
}
int srps_host_bilateral(const float* src, float* dst, int rows, int cols, float sc, float ss) {
    HOST_TRY({
        std::vector<float> a(src, src + (size_t)rows * cols), b;
        bilateral_filter(a, b, rows, cols, sc, ss);
        memcpy(dst, b.data(), b.size() * sizeof(float));
    })
}
int srps_host_resize_cubic(const float* src, int rows, int cols, float* dst, int out_rows, int out_cols) {
    HOST_TRY({
        std::vector<float> a(src, src + (size_t)rows * cols), b;
        resize_cubic(a, rows, cols, b, out_rows, out_cols);
        memcpy(dst, b.data(), b.size() * sizeof(float));
    })
}

// loaders: fill a DataHandler, report its sizes, copy its arrays out
struct srps_host_data { DataHandler dh; };
int srps_host_load(const char* dstype, const char* dsloc, srps_host_data** out) {
    HOST_TRY({
        srps_host_data* d = new srps_host_data();
        try {
            if (!strcmp(dstype, "matlab")) { MatFileDataHandler h; h.loadDataFromMatFiles(dsloc); d->dh = h; }
            else if (!strcmp(dstype, "images")) { ImageDataHandler h; h.loadDataFromImages(dsloc); d->dh = h; }
            else throw std::runtime_error(std::string("unknown dstype ") + dstype);
        } catch (...) { delete d; throw; }
        *out = d;
    })
}
int srps_host_data_dims(srps_host_data* d, int* I_h, int* I_w, int* I_c, int* I_n, int* z0_n, float* sf) {
    *I_h = d->dh.I_h; *I_w = d->dh.I_w; *I_c = d->dh.I_c; *I_n = d->dh.I_n; *z0_n = d->dh.z0_n; *sf = d->dh.sf;
    return 0;
}
int srps_host_data_copy(srps_host_data* d, float* I, float* mask, float* K, float* z0) {
    memcpy(I, d->dh.I.data(), d->dh.I.size() * sizeof(float)); memcpy(mask, d->dh.mask.data(), d->dh.mask.size() * sizeof(float));
    memcpy(K, d->dh.K.data(), 9 * sizeof(float)); memcpy(z0, d->dh.z0.data(), d->dh.z0.size() * sizeof(float));
    return 0;
}
void srps_host_data_free(srps_host_data* d) { delete d; }
// kind: 0 normals (N[4][P]), 1 albedo (rho[C][P]), 2 depth (z[P]); imask = HR linear indices of the masked pixels
int srps_host_write_view(int kind, const float* data, const int* imask, int P, int rows, int cols, int nchannels, float scale, const char* path) {
    HOST_TRY({
        std::vector<int> im(imask, imask + P);
        RgbImage v = kind == 0 ? normals_image(data, im, rows, cols) : kind == 1 ? albedo_image(data, im, rows, cols, nchannels) : depth_image(data, im, rows, cols);
        png_write_rgb8(path, scale == 1.f ? v : resize_bilinear(v, scale));
    })
}
int srps_host_write_mat_floats(const float* data, size_t n, const char* filename) { HOST_TRY({ write_MAT_floats(data, n, filename); }) }
}
