// MatIO.cpp -- minimal MAT-file level 5 reader / writer (numeric arrays, optional zlib-compressed
// elements).  Replaces the reference's use of matio 1.5.10 (Utilities.cpp:34-122, 159-199): the
// bundled Linux matio has no HDF5, so MAT5 is the only format it could ever read or write.
#include "MatIO.h"
#include <zlib.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>

namespace {

enum { miINT8 = 1, miUINT8 = 2, miINT16 = 3, miUINT16 = 4, miINT32 = 5, miUINT32 = 6, miSINGLE = 7, miDOUBLE = 9,
       miINT64 = 12, miUINT64 = 13, miMATRIX = 14, miCOMPRESSED = 15, miUTF8 = 16 };

struct Elem {
    uint32_t type = 0;
    const uint8_t* data = nullptr;
    size_t n = 0;
    size_t total = 0;      // bytes consumed including tag and padding
};

Elem read_elem(const uint8_t* p, size_t avail) {
    if (avail < 8) throw std::runtime_error("MAT5: truncated element tag");
    uint32_t w0, w1;
    memcpy(&w0, p, 4); memcpy(&w1, p + 4, 4);
    Elem e;
    if (w0 >> 16) {                         // small data element: bytes in the upper half of word 0
        e.type = w0 & 0xffff; e.n = w0 >> 16; e.data = p + 4; e.total = 8;
        if (e.n > 4) throw std::runtime_error("MAT5: bad small element");
    } else {
        e.type = w0; e.n = w1; e.data = p + 8;
        e.total = 8 + ((e.n + 7) & ~(size_t)7);
        if (e.type == miCOMPRESSED) e.total = 8 + e.n;      // compressed elements are not padded
        if (8 + e.n > avail) throw std::runtime_error("MAT5: truncated element data");
    }
    return e;
}

std::vector<uint8_t> inflate_all(const uint8_t* src, size_t n) {
    std::vector<uint8_t> out(n * 4 + 1024);
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (inflateInit(&zs) != Z_OK) throw std::runtime_error("MAT5: inflateInit failed");
    zs.next_in = const_cast<Bytef*>(src); zs.avail_in = (uInt)n;
    size_t have = 0;
    for (;;) {
        if (have == out.size()) out.resize(out.size() * 2);
        zs.next_out = out.data() + have;
        const size_t room = out.size() - have;
        zs.avail_out = (uInt)std::min<size_t>(room, 1u << 30);
        const uInt before = zs.avail_out;
        const int rc = inflate(&zs, Z_NO_FLUSH);
        have += before - zs.avail_out;
        if (rc == Z_STREAM_END) break;
        if (rc != Z_OK) { inflateEnd(&zs); throw std::runtime_error("MAT5: corrupt compressed element"); }
    }
    inflateEnd(&zs);
    out.resize(have);
    return out;
}

template <typename T>
void convert(const uint8_t* p, size_t count, std::vector<float>& out) {
    out.resize(count);
    for (size_t i = 0; i < count; ++i) { T v; memcpy(&v, p + i * sizeof(T), sizeof(T)); out[i] = (float)v; }
}

void numeric_to_float(const Elem& e, std::vector<float>& out) {
    switch (e.type) {
        case miINT8: convert<int8_t>(e.data, e.n, out); break;
        case miUINT8: convert<uint8_t>(e.data, e.n, out); break;
        case miINT16: convert<int16_t>(e.data, e.n / 2, out); break;
        case miUINT16: convert<uint16_t>(e.data, e.n / 2, out); break;
        case miINT32: convert<int32_t>(e.data, e.n / 4, out); break;
        case miUINT32: convert<uint32_t>(e.data, e.n / 4, out); break;
        case miSINGLE: convert<float>(e.data, e.n / 4, out); break;
        case miDOUBLE: convert<double>(e.data, e.n / 8, out); break;
        case miINT64: convert<int64_t>(e.data, e.n / 8, out); break;
        case miUINT64: convert<uint64_t>(e.data, e.n / 8, out); break;
        default: throw std::runtime_error("MAT5: unsupported numeric storage type " + std::to_string(e.type));
    }
}

void parse_matrix(const uint8_t* p, size_t n, std::map<std::string, MatVar>& vars) {
    size_t off = 0;
    Elem flags = read_elem(p + off, n - off); off += flags.total;
    if (flags.type != miUINT32 || flags.n < 8) throw std::runtime_error("MAT5: bad array flags");
    uint32_t f0; memcpy(&f0, flags.data, 4);
    const int cls = f0 & 0xff;
    Elem dims = read_elem(p + off, n - off); off += dims.total;
    Elem name = read_elem(p + off, n - off); off += name.total;
    MatVar v;
    v.name.assign((const char*)name.data, name.n);
    v.mx_class = cls;
    size_t count = 1;
    for (size_t i = 0; i + 4 <= dims.n; i += 4) { int32_t d; memcpy(&d, dims.data + i, 4); v.dims.push_back((size_t)d); count *= (size_t)d; }
    if (cls < 4 || cls > 15 || cls == 5) return;              // cell / struct / object / char / sparse: skipped
    if (count == 0) { vars[v.name] = v; return; }
    Elem real = read_elem(p + off, n - off);
    numeric_to_float(real, v.data);
    if (v.data.size() != count) throw std::runtime_error("MAT5: variable '" + v.name + "' has " + std::to_string(v.data.size()) + " values for " + std::to_string(count) + " elements");
    vars[v.name] = std::move(v);
}

void put32(std::vector<uint8_t>& b, uint32_t v) { uint8_t t[4]; memcpy(t, &v, 4); b.insert(b.end(), t, t + 4); }
void pad8(std::vector<uint8_t>& b) { while (b.size() % 8) b.push_back(0); }

void write_single_var(const char* filename, const char* varname, int mx_class, int mi_type, const void* data, size_t length, size_t elsize) {
    std::vector<uint8_t> body;
    put32(body, miUINT32); put32(body, 8); put32(body, (uint32_t)mx_class); put32(body, 0);        // array flags
    put32(body, miINT32); put32(body, 8); put32(body, (uint32_t)length); put32(body, 1);           // dims [len, 1]
    const size_t nl = strlen(varname);
    put32(body, miINT8); put32(body, (uint32_t)nl); body.insert(body.end(), varname, varname + nl); pad8(body);
    put32(body, (uint32_t)mi_type); put32(body, (uint32_t)(length * elsize));
    const uint8_t* d = (const uint8_t*)data;
    body.insert(body.end(), d, d + length * elsize); pad8(body);
    FILE* f = fopen(filename, "wb");
    if (!f) throw std::runtime_error(std::string("Error creating MAT file ") + filename);
    char hdr[128];
    memset(hdr, ' ', sizeof(hdr));
    const char* text = "MATLAB 5.0 MAT-file, Platform: GLNXA64, Created by: srps-hip";
    memcpy(hdr, text, strlen(text));
    memset(hdr + 116, 0, 8);
    hdr[124] = 0x00; hdr[125] = 0x01; hdr[126] = 'I'; hdr[127] = 'M';
    std::vector<uint8_t> tag;
    put32(tag, miMATRIX); put32(tag, (uint32_t)body.size());
    const bool ok = fwrite(hdr, 1, 128, f) == 128 && fwrite(tag.data(), 1, 8, f) == 8 && fwrite(body.data(), 1, body.size(), f) == body.size();
    fclose(f);
    if (!ok) throw std::runtime_error(std::string("Error writing MAT file ") + filename);
}

}  // namespace

std::map<std::string, MatVar> mat5_read(const std::string& path) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("Failed opening MAT file " + path);              // Utilities.cpp:165-168
    fseek(f, 0, SEEK_END);
    const long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> buf((size_t)std::max(0L, sz));
    const size_t got = fread(buf.data(), 1, buf.size(), f);
    fclose(f);
    if (got != buf.size() || buf.size() < 128) throw std::runtime_error("Failed reading MAT file " + path);
    if (memcmp(buf.data(), "MATLAB 5.0", 10) != 0) throw std::runtime_error(path + ": not a MAT-file level 5 (v7.3/HDF5 files are not supported)");
    if (!(buf[126] == 'I' && buf[127] == 'M')) throw std::runtime_error(path + ": big-endian MAT files are not supported");
    std::map<std::string, MatVar> vars;
    size_t off = 128;
    while (off + 8 <= buf.size()) {
        Elem e = read_elem(buf.data() + off, buf.size() - off);
        if (e.type == miCOMPRESSED) {
            std::vector<uint8_t> raw = inflate_all(e.data, e.n);
            Elem inner = read_elem(raw.data(), raw.size());
            if (inner.type == miMATRIX) parse_matrix(inner.data, inner.n, vars);
        } else if (e.type == miMATRIX) {
            parse_matrix(e.data, e.n, vars);
        }
        off += e.total;
    }
    return vars;
}

void mat5_write_single(const char* filename, const char* varname, const float* data, size_t length) {
    write_single_var(filename, varname, 7 /* mxSINGLE */, miSINGLE, data, length, 4);
}
void mat5_write_int32(const char* filename, const char* varname, const int* data, size_t length) {
    write_single_var(filename, varname, 12 /* mxINT32 */, miINT32, data, length, 4);
}
