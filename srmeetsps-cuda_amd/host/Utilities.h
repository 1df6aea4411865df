// Utilities.h -- host-side types of the SRPS command-line program on top of the C ABI (include/srps.h).
// Mirrors the surface of the reference's Utilities.h (SparseCOO :120-162, DataHandler :166-192,
// Timer :194-222, Preferences :224-230) without CUDA, OpenCV or matio: MAT files are read and
// written by host/MatIO.cpp, PNGs by host/PngIO.cpp.
#pragma once
#include <chrono>
#include <cstddef>
#include <stdexcept>
#include <string>
#include <vector>
#include "srps.h"

// reference: cusparse_check / cublas_check throw std::runtime_error (Utilities.cpp:21-31); the C ABI
// returns status codes, the facade turns them back into exceptions
inline void srps_check(int rc) {
    if (rc != SRPS_OK) throw std::runtime_error(std::string("SRPS ERROR ") + std::to_string(rc) + ": " + srps_last_error());
}

// Utilities.h:224-230 / Main.cpp:5-7
struct Preferences {
    static int blockX;
    static int blockY;
    static int deviceId;
    static bool writeOutputs;     // new: dump s/rho/z/N .mat after every pass (SRPS.cu:330-333)
    static bool writeImages;      // new: write the three imshow views (SRPS.cu:319-327) as PNG files
    static bool exclusiveDevice;  // new: the device is not shared (srps option "exclusive_device")
    static int numGpus;           // new: --gpus N: the images are sharded over devices deviceId .. deviceId + N - 1 of this node (one
                                  // thread and one context per device, RCCL all-reduces inside the library); the seam is the reference's
                                  // cudaSetDevice(Preferences::deviceId), SRPS.cu:88
    static bool partitionStrips;  // new: --partition strips: with --gpus N the depth CG is also cut into column strips over the devices (srps option "cg_partition")
    static bool forceSharded;     // new: --sharded: take the communicator path even on one GPU (a one-rank communicator; diagnostic)
    static std::string outDir;

private:
    Preferences() {}
};

// Utilities.h:120-162; std::vector storage instead of raw new[] (no manual freeMemory needed,
// kept as a no-op for source compatibility)
template <typename T>
struct SparseCOO {
    std::vector<int> row, col;
    std::vector<T> val;
    int n_row = 0, n_col = 0, n_nz = 0;
    SparseCOO() {}
    SparseCOO(int n_row_, int n_col_, int n_nz_) : row(n_nz_), col(n_nz_), val(n_nz_), n_row(n_row_), n_col(n_col_), n_nz(n_nz_) {}
    SparseCOO<T> operator+(const SparseCOO<T>& o) const {       // concatenation, as in the reference
        SparseCOO<T> r(n_row, n_col, n_nz + o.n_nz);
        std::copy(row.begin(), row.end(), r.row.begin()); std::copy(o.row.begin(), o.row.end(), r.row.begin() + n_nz);
        std::copy(col.begin(), col.end(), r.col.begin()); std::copy(o.col.begin(), o.col.end(), r.col.begin() + n_nz);
        std::copy(val.begin(), val.end(), r.val.begin()); std::copy(o.val.begin(), o.val.end(), r.val.begin() + n_nz);
        return r;
    }
    void freeMemory() { row.clear(); col.clear(); val.clear(); n_nz = 0; }
};

// Utilities.h:166-181. Column-major everywhere: I is h x w x c x n, mask h x w, K 3x3, z0 (h/sf) x (w/sf) x n.
struct DataHandler {
    std::vector<float> I;
    std::vector<unsigned char> I_u8;   // new: the same images as the bytes the image files hold (ImageDataHandler: I = byte / 255.f, Utilities.cpp:343),
                                       // same layout as I; empty when the data set did not come from 8-bit images.  SRPS::execute hands these to the
                                       // device instead of the floats (srps_problem.I_u8): a quarter of the bytes cross PCIe, the same floats are formed there
    int I_w = 0, I_h = 0, I_c = 0, I_n = 0;
    int z0_w = 0, z0_h = 0;
    std::vector<float> K;        // 9
    std::vector<float> mask;
    float sf = 1.f;
    std::vector<float> z0;
    int z0_n = 0;
    SparseCOO<float> D;          // only built on request (initializeDownsamplingMatrix); the solver does not need it
    void freeMemory();
    void initializeDownsamplingMatrix();       // Utilities.cpp:201-220
    void validate() const;
};

struct MatFileDataHandler : public DataHandler {
    void loadDataFromMatFiles(const char* filename);           // Utilities.cpp:159-199
};

struct ImageDataHandler : public DataHandler {
    void loadDataFromImages(const char* dataFolder);           // Utilities.cpp:349-395
};

// Utilities.h:194-222, but wall-clock (the reference's clock() measures process CPU time on Linux)
class Timer {
public:
    void start() { t0 = std::chrono::steady_clock::now(); running = true; }
    void end() { if (running) { sec = std::chrono::duration<float>(std::chrono::steady_clock::now() - t0).count(); running = false; } }
    float get() { if (running) end(); return sec; }

private:
    std::chrono::steady_clock::time_point t0;
    bool running = false;
    float sec = 0.f;
};

// result dumps of SRPS.cu:143, 250, 330-333 (variable "x", single, [len,1]) -- MAT5 (MatIO.cpp)
void write_MAT_floats(const float* data, size_t length, const char* filename);
void write_MAT_ints(const int* data, size_t length, const char* filename);
