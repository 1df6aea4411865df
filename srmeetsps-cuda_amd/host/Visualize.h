// Visualize.h -- headless replacement of the reference's imshow windows (SRPS.cu:319-327 with
// rho_as_opencv_mat / N_as_opencv_mat / z_as_opencv_mat, Utilities.cpp:242-320): the same images are
// written as 8-bit RGB PNG files instead of being shown (there is no display on a GPU box).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

struct RgbImage {
    int rows = 0, cols = 0;
    std::vector<float> px;       // row-major, interleaved R,G,B in [0,1]
};

// imask: HR linear column-major index of every masked pixel (SRPS.cu:157-162); rows = I_h, cols = I_w
RgbImage normals_image(const float* N, const std::vector<int>& imask, int rows, int cols);                 // Utilities.cpp:280-298
RgbImage albedo_image(const float* rho, const std::vector<int>& imask, int rows, int cols, int nchannels);  // Utilities.cpp:242-278
RgbImage depth_image(const float* z, const std::vector<int>& imask, int rows, int cols);                   // Utilities.cpp:300-320
RgbImage resize_bilinear(const RgbImage& src, float scale);                                                // cv::resize(..., scale, scale), INTER_LINEAR
void png_write_rgb8(const std::string& path, const RgbImage& im);
