// image3 on libmistereo: the main() of image3.cpp:920-1044 with its hot path (findOffset,
// calculateXYZ, the recursion) served by the C ABI of include/mi_stereo.h.  GDAL is not available
// to this build, so the two rasters are binary PGM files (P5, 8 or 16 bit); everything after the
// read is what a GDAL-reading host would do as well.
//
//   image3_mi left.pgm right.pgm [--xyz out.xyz] [--dispersion d.tif] [--min-size 32]
//             [--dense window dmin dmax disp.tif] [--devices 0,1,...]
// --devices: the recursion's tile batches and the dense correlation's row strips are shared out over
// the listed GPUs (mis_create_group); the outputs are the same files.
// Cameras are read from <left>.yaml and <right>.yaml (image3.cpp:1014-1023) when --xyz is given.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "mi_stereo.h"

static void die(const char* what, mis_ctx* ctx = nullptr)
{
    fprintf(stderr, "image3_mi: %s%s%s\n", what, ctx ? ": " : "", ctx ? mis_last_error(ctx) : "");
    exit(1);
}

// binary PGM; 16-bit samples are big-endian in the file
static bool read_pgm(const char* path, std::vector<unsigned char>& px, int& w, int& h, int& dtype)
{
    FILE* f = fopen(path, "rb");
    if (!f) return false;
    char magic[3] = { 0, 0, 0 };
    int maxv = 0;
    auto next_int = [&](int* v) {
        int c = fgetc(f);
        for (;;) {
            while (c == ' ' || c == '\n' || c == '\r' || c == '\t') c = fgetc(f);
            if (c == '#') { while (c != '\n' && c != EOF) c = fgetc(f); continue; }
            break;
        }
        if (c < '0' || c > '9') return false;
        int r = 0;
        while (c >= '0' && c <= '9') { r = r * 10 + (c - '0'); c = fgetc(f); }
        *v = r;                                   // the single whitespace after the last field is consumed here
        return true;
    };
    bool ok = fread(magic, 1, 2, f) == 2 && !strcmp(magic, "P5") && next_int(&w) && next_int(&h) && next_int(&maxv) &&
              w > 0 && h > 0 && maxv > 0 && maxv < 65536;
    if (ok) {
        const int bpp = maxv < 256 ? 1 : 2;
        dtype = bpp == 1 ? MIS_U8 : MIS_U16;
        px.resize((size_t)w * h * bpp);
        ok = fread(px.data(), 1, px.size(), f) == px.size();
        if (ok && bpp == 2)
            for (size_t i = 0; i + 1 < px.size(); i += 2) std::swap(px[i], px[i + 1]);
    }
    fclose(f);
    return ok;
}

int main(int argc, char** argv)
{
    const char *xyz_name = nullptr, *disp_name = nullptr, *dense_name = nullptr;
    int min_size = 32, window = 0, dmin = 0, dmax = 0;     // minSize default of image3.cpp:73
    std::vector<const char*> files;
    std::vector<int> devices;
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "--xyz") && i + 1 < argc) xyz_name = argv[++i];
        else if (!strcmp(argv[i], "--dispersion") && i + 1 < argc) disp_name = argv[++i];
        else if (!strcmp(argv[i], "--min-size") && i + 1 < argc) min_size = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--dense") && i + 4 < argc) {
            window = atoi(argv[++i]); dmin = atoi(argv[++i]); dmax = atoi(argv[++i]); dense_name = argv[++i];
        } else if (!strcmp(argv[i], "--devices") && i + 1 < argc) {
            for (const char* p = argv[++i]; *p;) {
                devices.push_back(atoi(p));
                while (*p && *p != ',') p++;
                if (*p == ',') p++;
            }
        } else files.push_back(argv[i]);
    }
    if (files.size() != 2) die("usage: image3_mi left.pgm right.pgm [--xyz f] [--dispersion f] [--min-size n] [--dense w dmin dmax f] [--devices 0,1,...]");

    std::vector<unsigned char> a, b;
    int w = 0, h = 0, w2 = 0, h2 = 0, t1 = 0, t2 = 0;
    if (!read_pgm(files[0], a, w, h, t1) || !read_pgm(files[1], b, w2, h2, t2)) die("Could not open source dataset");
    printf("Got image %d x %d x 1\nGot image %d x %d x 1\n", w, h, w2, h2);
    if (w != w2 || h != h2 || t1 != t2) die("Source dataset geometries should match");

    const size_t stride = (size_t)w * (t1 == MIS_U8 ? 1 : 2);
    mis_ctx* ctx = nullptr;
    mis_group* grp = nullptr;
    if (mis_create(devices.empty() ? 0 : devices[0], &ctx)) die(mis_last_error(nullptr));
    if (!devices.empty() && mis_create_group(devices.data(), (int)devices.size(), &grp)) die(mis_group_last_error(nullptr));

    // calcDisparityMap(bbox1, bbox2), image3.cpp:1032
    std::vector<mis_leaf> leaves(1 << 20);
    int n = 0, evaluated = 0;
    if (grp) {
        if (mis_group_upload_pair(grp, a.data(), b.data(), t1, w, h, stride)) die(mis_group_last_error(grp));
        if (mis_group_disparity_map(grp, w, h, min_size, leaves.data(), (int)leaves.size(), &n, &evaluated)) die(mis_group_last_error(grp));
    } else {
        if (mis_upload_pair(ctx, a.data(), b.data(), t1, w, h, stride)) die("upload", ctx);
        if (mis_disparity_map(ctx, w, h, min_size, leaves.data(), (int)leaves.size(), &n, &evaluated)) die("disparity map", ctx);
    }
    if (n > (int)leaves.size()) die("too many leaves");
    printf("%d tiles correlated, %d leaves\n", evaluated, n);

    if (xyz_name) {
        mis_camera c1, c2;
        if (mis_load_camera_yaml((std::string(files[0]) + ".yaml").c_str(), &c1) ||
            mis_load_camera_yaml((std::string(files[1]) + ".yaml").c_str(), &c2)) die("camera file missing or malformed");
        std::vector<int32_t> pts(4 * (size_t)n);
        std::vector<float> xyz(3 * (size_t)n), band(n);
        for (int i = 0; i < n; i++) {
            pts[4 * i] = leaves[i].u1; pts[4 * i + 1] = leaves[i].v1; pts[4 * i + 2] = leaves[i].u2; pts[4 * i + 3] = leaves[i].v2;
            // the band value written next to each point: the left pixel at (u1,v1), image3.cpp:838-841
            const size_t o = (size_t)leaves[i].v1 * w + leaves[i].u1;
            band[i] = t1 == MIS_U8 ? (float)a[o] : (float)((const unsigned short*)a.data())[o];
        }
        if (mis_triangulate(ctx, &c1, &c2, pts.data(), n, xyz.data())) die("triangulate", ctx);
        if (mis_write_xyz(xyz_name, xyz.data(), n, band.data(), 1, 0)) die("could not write the XYZ file");
    }
    if (disp_name) {
        std::vector<float> planes(3 * (size_t)w * h, 0.f);      // GDAL leaves unwritten pixels 0
        if (mis_rasterize_leaves(leaves.data(), n, w, h, planes.data()) ||
            mis_write_tiff_f32(disp_name, planes.data(), w, h, 3)) die("could not write the dispersion raster");
    }
    if (dense_name) {                                            // the dense mode image3.cpp:866-870 asks for
        std::vector<float> out(2 * (size_t)w * h);
        if (grp) {
            if (mis_group_ncc_dense(grp, a.data(), b.data(), t1, w, h, stride, window, dmin, dmax,
                                    out.data(), out.data() + (size_t)w * h, nullptr, nullptr)) die(mis_group_last_error(grp));
        } else if (mis_ncc_dense(ctx, window, dmin, dmax, out.data(), out.data() + (size_t)w * h, nullptr, nullptr)) die("dense NCC", ctx);
        if (mis_write_tiff_f32(dense_name, out.data(), w, h, 2)) die("could not write the dense disparity raster");
    }
    mis_destroy_group(grp);
    mis_destroy(ctx);
    return 0;
}
