// MatIO.h -- minimal MAT-file level 5 reader / writer (see MatIO.cpp)
#pragma once
#include <map>
#include <string>
#include <vector>

struct MatVar {
    std::string name;
    std::vector<size_t> dims;
    std::vector<float> data;     // numeric content converted to float (the reference casts to float too, Utilities.cpp:124-140)
    int mx_class = 0;
};

std::map<std::string, MatVar> mat5_read(const std::string& path);
void mat5_write_single(const char* filename, const char* varname, const float* data, size_t length);
void mat5_write_int32(const char* filename, const char* varname, const int* data, size_t length);
