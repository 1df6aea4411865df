// PngIO.h -- minimal PNG decoder (zlib inflate + unfiltering); replaces cv::imread in
// ImageDataHandler::loadDataFromImages (Utilities.cpp:349-395)
#pragma once
#include <cstdint>
#include <string>
#include <vector>

struct PngImage {
    int width = 0, height = 0;
    int channels = 0;            // 1 gray, 2 gray+alpha, 3 RGB, 4 RGBA (palette images are expanded to RGB)
    int bit_depth = 8;           // 8 or 16 (1/2/4-bit images are expanded to 8)
    std::vector<uint16_t> pix;   // row-major, interleaved, values 0..255 or 0..65535
};

PngImage png_read(const std::string& path);
// cv::imread(path) default flag: 8-bit, 3 channels (here RGB order), alpha dropped, gray replicated
std::vector<uint8_t> png_as_rgb8(const PngImage& im);
// cv::imread(path, GRAYSCALE): 8-bit gray (0.299 R + 0.587 G + 0.114 B for colour input)
std::vector<uint8_t> png_as_gray8(const PngImage& im);
// cv::imread(path, ANYDEPTH): gray, native depth
std::vector<uint16_t> png_as_gray_native(const PngImage& im);
