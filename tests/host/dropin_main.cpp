// dropin_main.cpp -- test driver for image-compression_amd/host/khugle_dropin.hpp (needs a HIP device).
//
// Plays the part of CImageProcessing::LoadBMPAndCompress / LoadComp (Main.cpp:180-334) around the drop-in entry
// points:  dropin_main <rows> <cols> <lvl> <rgb.raw> <out.comp> <coef.f64> <planes.f64>
//   rgb.raw     : planar u8 R, G, B
//   out.comp    : CompressImage + WriteAll
//   coef.f64    : the DWT coefficients CompressImage leaves in its input planes (Y, Cb, Cr)
//   planes.f64  : ReadAll + DecompressImage outputs (Y, Cb, Cr)
#include <cstdio>
#include <cstdlib>

#include "khugle_dropin.hpp"

namespace {

double** dmatrix(int h, int w)  // the application's plane type: one allocation per row (KhuGleBase.cpp:188-199)
{
    double** m = new double*[h];
    for (int y = 0; y < h; ++y) m[y] = new double[w]();
    return m;
}

void dump(std::FILE* f, double** m, int h, int w)
{
    for (int y = 0; y < h; ++y) std::fwrite(m[y], sizeof(double), static_cast<size_t>(w), f);
}

}  // namespace

int main(int argc, char** argv)
{
    if (argc != 8) return 2;
    const int rows = std::atoi(argv[1]), cols = std::atoi(argv[2]), lvl = std::atoi(argv[3]);
    const size_t n = static_cast<size_t>(rows) * cols;
    std::vector<uint8_t> rgb(3 * n);
    std::FILE* f = std::fopen(argv[4], "rb");
    if (!f || std::fread(rgb.data(), 1, rgb.size(), f) != rgb.size()) return 3;
    std::fclose(f);
    const uint8_t *R = rgb.data(), *G = R + n, *B = G + n;
    double **Y = dmatrix(rows, cols), **Cb = dmatrix(rows / 2, cols / 2), **Cr = dmatrix(rows / 2, cols / 2);
    // Main.cpp:196-206 + Colorspace.h:5-23: products and sums left to right
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            const double r = R[static_cast<size_t>(y) * cols + x], g = G[static_cast<size_t>(y) * cols + x],
                         b = B[static_cast<size_t>(y) * cols + x];
            Y[y][x] = r * 0.299 + g * 0.587 + b * 0.114;
            if (y % 2 == 0 && x % 2 == 0) {
                Cb[y / 2][x / 2] = r * -0.16874 + g * -0.33126 + b * 0.5;
                Cr[y / 2][x / 2] = r * 0.5 + g * -0.41869 + b * -0.08131;
            }
        }
    try {
        std::unique_ptr<khugle_hip::CompResult> res;
        std::vector<int> gui;  // stands for the application's image vector
        std::string s1, s2;
        khugle_hip::CompressImage(res, Y, Cb, Cr, rows, cols, lvl, gui, s1, s2);
        khugle_hip::WriteAll(argv[5], res.get());
        f = std::fopen(argv[6], "wb");
        dump(f, Y, rows, cols); dump(f, Cb, rows / 2, cols / 2); dump(f, Cr, rows / 2, cols / 2);
        std::fclose(f);
        double **OY = dmatrix(rows, cols), **OCb = dmatrix(rows / 2, cols / 2), **OCr = dmatrix(rows / 2, cols / 2);
        std::string d1, d2;
        khugle_hip::DecompressImage(argv[5], OY, OCb, OCr, gui, d1, d2);
        if (d1 != s1 || d2 != s2) throw std::runtime_error("status lines of the two directions differ");
        std::printf("%s\n%s\n", s1.c_str(), s2.c_str());
        f = std::fopen(argv[7], "wb");
        dump(f, OY, rows, cols); dump(f, OCb, rows / 2, cols / 2); dump(f, OCr, rows / 2, cols / 2);
        std::fclose(f);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "dropin_main: %s\n", e.what());
        return 1;
    }
    return 0;
}
