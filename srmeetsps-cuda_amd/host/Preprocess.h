// Preprocess.h -- CPU depth pre-processing of SRPS::execute (SRPS.cu:117-149): channel mean with
// zero flagging, inpainting of the flagged pixels, max-normalised bilateral smoothing and cubic
// up-sampling.  The reference calls OpenCV 3.3 (cv::inpaint TELEA r=16, cv::bilateralFilter(-1,2,2),
// cv::resize INTER_CUBIC); OpenCV is not available here, so these are re-implementations to the
// published definitions.  PARITY UNPINNED against OpenCV's exact arithmetic.
#pragma once
#include <cstdint>
#include <vector>

// images are row-major `rows x cols` floats.  NB: the reference views its column-major (h x w) arrays
// as row-major (w x h) cv::Mat (SRPS.cu:130-132, 148), i.e. it processes the transposed image; callers
// here do the same by passing rows = z0_w, cols = z0_h.
void mean_across_channels_cpu(const float* z0, int n_pix, int nc, std::vector<float>& mean, std::vector<uint8_t>& flag);  // dc.cu:95-110
void inpaint_telea(std::vector<float>& img, const std::vector<uint8_t>& flag, int rows, int cols, int radius);            // SRPS.cu:133
void bilateral_filter(const std::vector<float>& src, std::vector<float>& dst, int rows, int cols, float sigma_color, float sigma_space);  // SRPS.cu:139 (d = -1)
void resize_cubic(const std::vector<float>& src, int rows, int cols, std::vector<float>& dst, int out_rows, int out_cols);  // SRPS.cu:149

// the whole chain: z0 [z0_n][z0_h*z0_w] column-major -> zs (smoothed LR, column-major) and z_full (HR, column-major)
void preprocess_depth(const float* z0, int z0_h, int z0_w, int z0_n, int I_h, int I_w,
                      std::vector<float>& zs, std::vector<float>& z_full);
