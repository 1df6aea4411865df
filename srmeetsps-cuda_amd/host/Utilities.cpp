// Utilities.cpp -- DataHandler and the two loaders (reference: Utilities.cpp:124-220, 322-395)
#include "Utilities.h"
#include <dirent.h>
#include <algorithm>
#include <cmath>
#include <fstream>
#include <sstream>
#include "MatIO.h"
#include "PngIO.h"

int Preferences::blockX = 256;          // Main.cpp:5-7
int Preferences::blockY = 4;
int Preferences::deviceId = 0;
bool Preferences::writeOutputs = true;
bool Preferences::writeImages = false;
bool Preferences::exclusiveDevice = false;
int Preferences::numGpus = 1;
bool Preferences::forceSharded = false;
bool Preferences::partitionStrips = false;
std::string Preferences::outDir = ".";

void DataHandler::freeMemory() {
    I.clear(); I_u8.clear(); K.clear(); mask.clear(); z0.clear(); D.freeMemory();
}

void DataHandler::validate() const {
    if (I_h <= 0 || I_w <= 0 || I_c <= 0 || I_n <= 0) throw std::runtime_error("DataHandler: empty image set");
    const int isf = (int)sf;
    if ((float)isf != sf || isf < 1 || I_h % isf || I_w % isf) throw std::runtime_error("DataHandler: sf must be an integer dividing the image size");
    if (I.size() != (size_t)I_h * I_w * I_c * I_n) throw std::runtime_error("DataHandler: I has the wrong size");
    if (mask.size() != (size_t)I_h * I_w) throw std::runtime_error("DataHandler: mask has the wrong size");
    if (K.size() != 9) throw std::runtime_error("DataHandler: K must be 3x3");
    if (z0.size() != (size_t)(I_h / isf) * (I_w / isf) * z0_n || z0_n < 1) throw std::runtime_error("DataHandler: z0 has the wrong size");
}

// Utilities.cpp:201-220 (not needed by the solver, which never forms D; kept for API parity and tests)
void DataHandler::initializeDownsamplingMatrix() {
    const int isf = (int)sf, per = isf * isf;
    D = SparseCOO<float>(I_h * I_w / per, I_h * I_w, (I_h * I_w / per) * per);
    const int hs = I_h / isf;
    for (int i = 0; i < D.n_row; ++i)
        for (int j = 0; j < isf; ++j)
            for (int k = 0; k < isf; ++k) {
                const int e = i * per + j * isf + k;
                D.row[e] = i;
                D.val[e] = 1.f / (sf * sf);
                D.col[e] = (i / hs) * I_h * isf + (i % hs) * isf + j * I_h + k;          // Utilities.cpp:216
            }
}

// ---- MAT file: variables I (h x w x c x n), K (3x3), mask (h x w), sf, z0 ((h/sf) x (w/sf) [x n]) ----
void MatFileDataHandler::loadDataFromMatFiles(const char* filename) {
    freeMemory();
    std::map<std::string, MatVar> v = mat5_read(filename);
    auto need = [&](const char* name) -> MatVar& {
        auto it = v.find(name);
        if (it == v.end()) throw std::runtime_error(std::string("Failed reading MAT file: variable '") + name + "' not found");   // Utilities.cpp:37-41
        return it->second;
    };
    MatVar& mI = need("I");
    if (mI.dims.size() < 2) throw std::runtime_error("MAT file: I must be h x w x c x n");
    I_h = (int)mI.dims[0]; I_w = (int)mI.dims[1];
    I_c = mI.dims.size() > 2 ? (int)mI.dims[2] : 1;
    I_n = mI.dims.size() > 3 ? (int)mI.dims[3] : 1;                                  // Utilities.cpp:171
    I = std::move(mI.data);
    K = need("K").data;
    mask = need("mask").data;                                                         // uint8/logical -> float, Utilities.cpp:181-183
    sf = need("sf").data.at(0);
    MatVar& mz = need("z0");
    z0_n = mz.dims.size() > 2 ? (int)mz.dims[2] : 1;                                 // Utilities.cpp:191
    z0_h = (int)mz.dims[0]; z0_w = (int)mz.dims[1];
    z0 = std::move(mz.data);
    validate();
}

// ---- image folder: RGB/*.png, Depth/*.png (16 bit), mask.png, K.txt ---------------------------------
static std::vector<std::string> glob_sorted(const std::string& dir) {           // cv::glob: lexicographic order
    std::vector<std::string> out;
    DIR* d = opendir(dir.c_str());
    if (!d) throw std::runtime_error("cannot open directory " + dir);
    while (dirent* e = readdir(d)) {
        std::string n = e->d_name;
        if (n == "." || n == "..") continue;
        out.push_back(dir + "/" + n);
    }
    closedir(d);
    std::sort(out.begin(), out.end());
    if (out.empty()) throw std::runtime_error("no files in " + dir);
    return out;
}

void ImageDataHandler::loadDataFromImages(const char* dataFolder) {
    freeMemory();
    const std::string root(dataFolder);
    std::vector<std::string> files = glob_sorted(root + "/RGB");                    // Utilities.cpp:352
    PngImage first = png_read(files[0]);
    I_n = (int)files.size(); I_w = first.width; I_h = first.height; I_c = 3;        // imread default: 3 channels
    I.resize((size_t)I_h * I_w * I_c * I_n);
    I_u8.resize(I.size());
    for (int n = 0; n < I_n; ++n) {
        PngImage im = n == 0 ? first : png_read(files[n]);
        if (im.width != I_w || im.height != I_h) throw std::runtime_error(files[n] + ": image size differs");
        std::vector<uint8_t> rgb = png_as_rgb8(im);
        float* dst = I.data() + (size_t)n * I_w * I_h * I_c;
        unsigned char* dst8 = I_u8.data() + (size_t)n * I_w * I_h * I_c;
        for (int c = 0; c < 3; ++c)                                                   // plane 0 = R, 1 = G, 2 = B (Utilities.cpp:343)
            for (int i = 0; i < I_h; ++i)
                for (int j = 0; j < I_w; ++j) {
                    const uint8_t b = rgb[((size_t)i * I_w + j) * 3 + c];
                    dst8[i + (size_t)j * I_h + (size_t)c * I_h * I_w] = b;            // what the device gets
                    dst[i + (size_t)j * I_h + (size_t)c * I_h * I_w] = b / 255.f;     // the DataHandler's float view of it
                }
    }
    std::ifstream fk(root + "/K.txt");
    if (!fk) throw std::runtime_error("cannot open " + root + "/K.txt");
    K.assign(9, 0.f);
    std::string line, val;
    for (int i = 0; i < 3; ++i) {                                                     // Utilities.cpp:366-375
        std::getline(fk, line);
        std::istringstream tk(line);
        for (int j = 0; j < 3; ++j) { std::getline(tk, val, ','); K[i + 3 * j] = std::stof(val); }
    }
    std::getline(fk, line);
    float min_z, max_z;
    {
        std::istringstream tk(line);
        std::getline(tk, val, ','); sf = std::stof(val);
        std::getline(tk, val, ','); min_z = std::stof(val);
        std::getline(tk, val); max_z = std::stof(val);                                // Utilities.cpp:376-383
    }
    {
        PngImage pm = png_read(root + "/mask.png");
        if (pm.width != I_w || pm.height != I_h) throw std::runtime_error("mask.png: size differs from the images");
        std::vector<uint8_t> g = png_as_gray8(pm);
        mask.resize((size_t)I_h * I_w);
        for (int i = 0; i < I_h; ++i)
            for (int j = 0; j < I_w; ++j) mask[i + (size_t)j * I_h] = g[(size_t)i * I_w + j] / 255.f;   // Utilities.cpp:330, 385
    }
    files = glob_sorted(root + "/Depth");
    z0_n = (int)files.size();
    z0_h = (int)(I_h / sf); z0_w = (int)(I_w / sf);
    z0.resize((size_t)z0_h * z0_w * z0_n);
    for (int n = 0; n < z0_n; ++n) {
        PngImage pd = png_read(files[n]);
        if (pd.width != z0_w || pd.height != z0_h) throw std::runtime_error(files[n] + ": depth size is not image size / sf");
        std::vector<uint16_t> g = png_as_gray_native(pd);
        const float q = pd.bit_depth == 16 ? 65535.f : 255.f;
        for (int i = 0; i < z0_h; ++i)
            for (int j = 0; j < z0_w; ++j)
                z0[(size_t)n * z0_h * z0_w + i + (size_t)j * z0_h] = min_z + (g[(size_t)i * z0_w + j] / q) * (max_z - min_z);   // Utilities.cpp:330, 392
    }
    validate();
}

void write_MAT_floats(const float* data, size_t length, const char* filename) { mat5_write_single(filename, "x", data, length); }
void write_MAT_ints(const int* data, size_t length, const char* filename) { mat5_write_int32(filename, "x", data, length); }
