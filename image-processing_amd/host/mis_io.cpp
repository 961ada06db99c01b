// Host-side file formats either side of the hot path: the camera files image3.cpp reads
// (OpenCV FileStorage YAML, image3.cpp:1014-1030) and the 3-band Float32 band-interleaved TIFF it
// writes through GDAL (image3.cpp:872-893).  Plain host C++ above the C ABI: no HIP here.
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "mi_stereo.h"

namespace {

// One "!!opencv-matrix" node: rows, cols, dt and the flow sequence after "data:" (which OpenCV
// wraps over several lines).  Returns false when the key is absent or malformed.
bool read_matrix(const std::string& text, const char* key, int want_rows, int want_cols, float* out)
{
    // top-level key: at the start of a line, followed by ':'
    size_t pos = 0;
    const std::string k = std::string(key) + ":";
    for (;;) {
        pos = text.find(k, pos);
        if (pos == std::string::npos) return false;
        if (pos == 0 || text[pos - 1] == '\n') break;
        pos += k.size();
    }
    // the node ends where the next top-level key starts (a line that does not begin with a blank)
    size_t end = pos;
    for (;;) {
        end = text.find('\n', end);
        if (end == std::string::npos) { end = text.size(); break; }
        end++;
        if (end >= text.size()) break;
        if (!isspace((unsigned char)text[end])) break;
    }
    const std::string node = text.substr(pos, end - pos);
    auto int_field = [&](const char* name, int* v) {
        size_t p = node.find(std::string(name) + ":");
        if (p == std::string::npos) return false;
        *v = atoi(node.c_str() + p + strlen(name) + 1);
        return true;
    };
    int rows = 0, cols = 0;
    if (!int_field("rows", &rows) || !int_field("cols", &cols)) return false;
    if (rows * cols != want_rows * want_cols) return false;       // T may be 3x1 or 1x3
    size_t d = node.find("data:");
    if (d == std::string::npos) return false;
    size_t lb = node.find('[', d), rb = node.find(']', d);
    if (lb == std::string::npos || rb == std::string::npos || rb < lb) return false;
    std::string body = node.substr(lb + 1, rb - lb - 1);
    for (char& c : body) if (c == ',' || c == '\n' || c == '\r' || c == '\t') c = ' ';
    const char* p = body.c_str();
    int n = 0;
    while (*p) {
        while (*p == ' ') p++;
        if (!*p) break;
        double v;
        if (!strncmp(p, ".Inf", 4) || !strncmp(p, ".inf", 4)) { v = INFINITY; p += 4; }
        else if (!strncmp(p, "-.Inf", 5) || !strncmp(p, "-.inf", 5)) { v = -INFINITY; p += 5; }
        else if (!strncmp(p, ".Nan", 4) || !strncmp(p, ".NaN", 4) || !strncmp(p, ".nan", 4)) { v = NAN; p += 4; }
        else {
            char* e = nullptr;
            v = strtod(p, &e);
            if (e == p) return false;
            p = e;
        }
        if (n >= want_rows * want_cols) return false;
        out[n++] = (float)v;                     // Mat::assignTo(..., CV_32F), image3.cpp:1025-1030
    }
    return n == want_rows * want_cols;
}

void put16(std::vector<unsigned char>& b, unsigned v) { b.push_back(v & 255); b.push_back((v >> 8) & 255); }
void put32(std::vector<unsigned char>& b, unsigned long long v) { for (int i = 0; i < 4; i++) b.push_back((v >> (8 * i)) & 255); }
void put64(std::vector<unsigned char>& b, unsigned long long v) { for (int i = 0; i < 8; i++) b.push_back((v >> (8 * i)) & 255); }

}  // namespace

extern "C" {

int mis_load_camera_yaml(const char* path, mis_camera* cam)
{
    if (!path || !cam) return MIS_ERR_ARG;
    FILE* f = fopen(path, "rb");
    if (!f) return MIS_ERR_STATE;
    std::string text;
    char buf[4096];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) text.append(buf, n);
    fclose(f);
    if (!read_matrix(text, "A", 3, 3, cam->A) || !read_matrix(text, "R", 3, 3, cam->R) ||
        !read_matrix(text, "T", 3, 1, cam->T))
        return MIS_ERR_ARG;
    return MIS_OK;
}

int mis_write_tiff_f32(const char* path, const float* planes, int width, int height, int bands)
{
    if (!path || !planes || width <= 0 || height <= 0 || bands <= 0 || bands > 64) return MIS_ERR_ARG;
    const unsigned long long plane = (unsigned long long)width * height * 4;
    const bool big = plane * bands + 4096 > 0xFFFF0000ull;        // BigTIFF beyond 4 GiB
    // one strip per band, pixel data first, directory at the end
    const unsigned long long data0 = big ? 16 : 8;
    std::vector<unsigned char> tail;                               // out-of-line tag values, then the IFD
    const unsigned long long tail0 = data0 + plane * bands;
    auto tail_pos = [&]() { return tail0 + tail.size(); };

    struct Tag { unsigned id, type; unsigned long long count, value; };
    std::vector<Tag> tags;
    // arrays with one entry per band
    auto per_band = [&](unsigned id, unsigned type, int size, auto value_of) {
        if ((unsigned long long)size * bands <= (big ? 8u : 4u)) {
            unsigned long long packed = 0;
            for (int b = 0; b < bands; b++) packed |= (unsigned long long)value_of(b) << (8 * size * b);
            tags.push_back({ id, type, (unsigned long long)bands, packed });
        } else {
            const unsigned long long at = tail_pos();
            for (int b = 0; b < bands; b++) {
                if (size == 2) put16(tail, (unsigned)value_of(b));
                else if (size == 4) put32(tail, value_of(b));
                else put64(tail, value_of(b));
            }
            tags.push_back({ id, type, (unsigned long long)bands, at });
        }
    };
    tags.push_back({ 256, 3, 1, (unsigned long long)width });                     // ImageWidth (SHORT or LONG)
    tags.push_back({ 257, 3, 1, (unsigned long long)height });
    if (width > 65535) tags[0].type = 4;
    if (height > 65535) tags[1].type = 4;
    per_band(258, 3, 2, [](int) { return 32ull; });                               // BitsPerSample
    tags.push_back({ 259, 3, 1, 1 });                                             // no compression
    tags.push_back({ 262, 3, 1, 1 });                                             // MinIsBlack
    if (big) per_band(273, 16, 8, [&](int b) { return data0 + plane * b; });      // StripOffsets
    else per_band(273, 4, 4, [&](int b) { return data0 + plane * b; });
    tags.push_back({ 277, 3, 1, (unsigned long long)bands });                     // SamplesPerPixel
    tags.push_back({ 278, height > 65535 ? 4u : 3u, 1, (unsigned long long)height });   // RowsPerStrip
    if (big) per_band(279, 16, 8, [&](int) { return plane; });                    // StripByteCounts
    else per_band(279, 4, 4, [&](int) { return plane; });
    tags.push_back({ 284, 3, 1, 2 });                                             // PlanarConfiguration = separate
    if (bands > 1) {                                                              // ExtraSamples: unspecified
        const int extra = bands - 1;
        if ((unsigned)extra * 2 <= (big ? 8u : 4u)) tags.push_back({ 338, 3, (unsigned long long)extra, 0 });
        else {
            const unsigned long long at = tail_pos();
            for (int b = 0; b < extra; b++) put16(tail, 0);
            tags.push_back({ 338, 3, (unsigned long long)extra, at });
        }
    }
    per_band(339, 3, 2, [](int) { return 3ull; });                                // SampleFormat = IEEE float
    if (tail.size() & 1) tail.push_back(0);
    const unsigned long long ifd_at = tail_pos();
    if (big) {
        put64(tail, tags.size());
        for (const Tag& t : tags) { put16(tail, t.id); put16(tail, t.type); put64(tail, t.count); put64(tail, t.value); }
        put64(tail, 0);
    } else {
        put16(tail, (unsigned)tags.size());
        for (const Tag& t : tags) { put16(tail, t.id); put16(tail, t.type); put32(tail, t.count); put32(tail, t.value); }
        put32(tail, 0);
    }
    std::vector<unsigned char> head;
    head.push_back('I'); head.push_back('I');
    if (big) { put16(head, 43); put16(head, 8); put16(head, 0); put64(head, ifd_at); }
    else { put16(head, 42); put32(head, ifd_at); }

    FILE* f = fopen(path, "wb");
    if (!f) return MIS_ERR_STATE;
    bool ok = fwrite(head.data(), 1, head.size(), f) == head.size();
    const size_t total = (size_t)width * height * bands;
    ok = ok && fwrite(planes, sizeof(float), total, f) == total;   // little-endian host assumed (x86-64)
    ok = ok && fwrite(tail.data(), 1, tail.size(), f) == tail.size();
    ok = (fclose(f) == 0) && ok;
    return ok ? MIS_OK : MIS_ERR_STATE;
}

}  // extern "C"
