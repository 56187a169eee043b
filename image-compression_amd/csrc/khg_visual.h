// khg_visual.h -- the pictures the application shows beside the result of CompressImage / DecompressImage
// (Compress.h:116-141, 245-288, 330-391; Decompress.h:128-262): the coefficient planes pushed through the colour
// conversion, the quantiser's step sizes as shades of red, and the three stream lengths as bars.  One thread per pixel.
#pragma once
#include "khg_common.h"
#include "khg_decode.h"
#include "khg_inverse.h"

// Sub-band of position (y, x) of an h x w coefficient plane in the order of Compress.h:154-160:
// LL1, LH1, HL1, HH1 (level 2, in the top left quadrant), LH, HL, HH (level 1).
KHG_HOST_DEV int khg_vis_band(int y, int x, int h, int w)
{
    if (y < h / 2 && x < w / 2) return (y < h / 4 ? 0 : 2) + (x < w / 4 ? 0 : 1);
    if (y < h / 2) return 4;
    return x < w / 2 ? 5 : 6;
}

// ---- the coefficient planes as a picture: YCbCr2RGB (Colorspace.h:25-36) on whatever the planes hold, then the
// conversion to unsigned char (Compress.h:130-139, Decompress.h:252-261)
struct KhgVisYccArgs {
    int rows, cols;
    // source A: f64 planes (Y rows x cols; Cb, Cr rows/2 x cols/2): the DWT coefficients of CompressImage
    const double *y, *cb, *cr;
    // source B (y == null): the decoded symbols times the quantiser's divisor (Decompress.h:56-126), preset from hdr
    const int16_t* sym[3];
    const KhgDecHeader* hdr;
    KhgQuant q[2][3];  // [luma, chroma][preset]
    uint8_t *r, *g, *b;
};

KHG_DEV void khg_vis_ycc_body(const KhgVisYccArgs& a, uint32_t bid)
{
    if (!a.y && !a.hdr->ok) return;
    KHG_FOR_THREADS(256)
    {
        const size_t p = (size_t)bid * 256 + (size_t)tid;
        if (p < (size_t)a.rows * (size_t)a.cols) {
            const int i = (int)(p / (size_t)a.cols), j = (int)(p - (size_t)i * (size_t)a.cols);
            const int ci = i / 2, cj = j / 2, cw = a.cols / 2;
            const size_t c = (size_t)ci * (size_t)cw + (size_t)cj;
            double Y, Cb, Cr;
            if (a.y) {
                Y = a.y[p];
                Cb = a.cb[c];
                Cr = a.cr[c];
            } else {
                const int lvl = a.hdr->lvl;
                const double dy = khg_divisor(a.q[0][lvl], khg_vis_band(i, j, a.rows, a.cols), j, i, lvl);
                const double dc = khg_divisor(a.q[1][lvl], khg_vis_band(ci, cj, a.rows / 2, a.cols / 2), cj, ci, lvl);
                Y = (double)a.sym[0][p] * dy;   // Decompress.h:105
                Cb = (double)a.sym[1][c] * dc;
                Cr = (double)a.sym[2][c] * dc;
            }
            // Colorspace.h:31-33, products and sums left to right; clamp and truncation as khg_clamp_u8 says
            a.r[p] = (uint8_t)khg_clamp_u8(Y * 1 + Cr * 1.402);
            a.g[p] = (uint8_t)khg_clamp_u8(Y * 1 + Cb * -0.34414 + Cr * -0.71414);
            a.b[p] = (uint8_t)khg_clamp_u8(Y * 1 + Cb * 1.772);
        }
    }
}

// ---- the quantiser's step sizes in red (Compress.h:265-288, Decompress.h:207-232): presets 1 and 2 show the
// per-position divisor, (d - min) / max * 255; preset 0 the seven band steps the same way.  lo / hi are that minimum
// and maximum (khg_vis_step_range).
struct KhgVisStepArgs {
    int rows, cols, lvl;
    KhgQuant q;  // luma, preset lvl
    double lo, hi;
    uint8_t *r, *g, *b;
};

KHG_DEV void khg_vis_step_body(const KhgVisStepArgs& a, uint32_t bid)
{
    KHG_FOR_THREADS(256)
    {
        const size_t p = (size_t)bid * 256 + (size_t)tid;
        if (p < (size_t)a.rows * (size_t)a.cols) {
            const int y = (int)(p / (size_t)a.cols), x = (int)(p - (size_t)y * (size_t)a.cols);
            const int band = khg_vis_band(y, x, a.rows, a.cols);
            double n = (khg_divisor(a.q, band, x, y, a.lvl) - a.lo) / a.hi;  // at preset 0 the divisor is the band's step
            n *= 255;
            a.r[p] = (uint8_t)(uint32_t)n;  // 0 <= n < 255
            a.g[p] = 0;
            a.b[p] = 0;
        }
    }
}

// The extremes the step picture is scaled by, and the status line prints (Compress.h:259-263, 294-296): of the seven
// band steps at preset 0, of the divisor over the whole luma plane otherwise.  Within a band the divisor grows with
// x + y (a > 0; a product and a sum of non-negative doubles are monotone), so its extremes sit at the band's first and
// last position, evaluated with the expression the quantiser uses.
static inline void khg_vis_step_range(int rows, int cols, int lvl, const KhgQuant& q, double* lo, double* hi)
{
    const int h = rows, w = cols;
    const int map[7][4] = {{0, h / 4, 0, w / 4},         {0, h / 4, w / 4, w / 2}, {h / 4, h / 2, 0, w / 4},
                           {h / 4, h / 2, w / 4, w / 2}, {0, h / 2, w / 2, w},     {h / 2, h, 0, w / 2},
                           {h / 2, h, w / 2, w}};
    double mn = 0, mx = 0;
    for (int band = 0; band < 7; ++band) {
        const double first = khg_divisor(q, band, map[band][2], map[band][0], lvl);
        const double last = khg_divisor(q, band, map[band][3] - 1, map[band][1] - 1, lvl);
        if (band == 0 || first < mn) mn = first;
        if (band == 0 || last > mx) mx = last;
        if (last < mn) mn = last;
        if (first > mx) mx = first;
    }
    *lo = mn;
    *hi = mx;
}

// ---- the three stream lengths as bars (Compress.h:330-391, Decompress.h:128-188): stream c covers the first
// len[c] / rows image rows, the last of them only up to column len[c] % rows, and adds its colour there in unsigned
// char arithmetic (so the sums wrap).  The reference divides by the image height for both and lets the last row's
// end run to that remainder even where it exceeds the width, or the rows exceed the image (it writes out of bounds
// then); here positions outside the picture are simply not drawn.
struct KhgVisBarArgs {
    int rows, cols;
    long long len[3];  // bytes of the Y, Cb, Cr streams
    uint8_t *r, *g, *b;
};

KHG_DEV void khg_vis_bar_body(const KhgVisBarArgs& a, uint32_t bid)
{
    KHG_FOR_THREADS(256)
    {
        const size_t p = (size_t)bid * 256 + (size_t)tid;
        if (p < (size_t)a.rows * (size_t)a.cols) {
            const int y = (int)(p / (size_t)a.cols), x = (int)(p - (size_t)y * (size_t)a.cols);
            const uint8_t add[3][3] = {{76, 149, 29}, {(uint8_t)-43, (uint8_t)-84, 127}, {127, (uint8_t)-106, (uint8_t)-20}};
            uint8_t v[3] = {0, 0, 0};
            for (int c = 0; c < 3; ++c) {
                const long long full = a.len[c] / a.rows, rem = a.len[c] % a.rows;
                if ((long long)y < full && (long long)x < ((long long)y == full - 1 ? rem : (long long)a.cols)) {
                    v[0] = (uint8_t)(v[0] + add[c][0]);
                    v[1] = (uint8_t)(v[1] + add[c][1]);
                    v[2] = (uint8_t)(v[2] + add[c][2]);
                }
            }
            a.r[p] = v[0];
            a.g[p] = v[1];
            a.b[p] = v[2];
        }
    }
}

// ---- the thumbnails the application draws its pictures as (DownSample.h:5-37, called for 256 x 256 from
// Main.cpp:59-61): bilinear interpolation in single precision, operation for operation.  Where the reference divides
// by zero (a sample position on the last row or column of the source: x2 == x1) its result is NaN or infinite and
// the conversion to unsigned char is undefined; this does what the reference's x86-64 build does there
// (cvttss2si gives INT_MIN, whose low byte is 0).
struct KhgDownArgs {
    const uint8_t* src;
    int src_h, src_w;
    uint8_t* dst;
    int dst_h, dst_w;
};

// float -> unsigned char as the reference's build converts it: truncate to a 32-bit int (INT_MIN when that is
// impossible), keep the low byte.
KHG_DEV uint8_t khg_f32_to_u8(float v)
{
    if (!(v > -2147483904.0f && v < 2147483648.0f)) return 0;  // NaN, infinities, out of int range
    return (uint8_t)(uint32_t)(int32_t)v;
}

#ifdef KHG_SIM
KHG_DEV float khg_div_f32(float a, float b) { return a / b; }
#else
KHG_DEV float khg_div_f32(float a, float b) { return __fdiv_rn(a, b); }  // correctly rounded, whatever the build's default
#endif

KHG_DEV void khg_downsample_body(const KhgDownArgs& a, uint32_t bid)
{
    KHG_FOR_THREADS(256)
    {
        const size_t p = (size_t)bid * 256 + (size_t)tid;
        if (p < (size_t)a.dst_h * (size_t)a.dst_w) {
            const int yy = (int)(p / (size_t)a.dst_w), xx = (int)(p - (size_t)yy * (size_t)a.dst_w);
            const float scaleX = (float)a.src_w / (float)a.dst_w, scaleY = (float)a.src_h / (float)a.dst_h;  // DownSample.h:29-30
            const float x = (float)xx * scaleX, y = (float)yy * scaleY;                                  // :34
            int x1 = (int)__builtin_floorf(x), y1 = (int)__builtin_floorf(y);                            // :7-8
            x1 = x1 < a.src_w - 1 ? x1 : a.src_w - 1;  // never taken for the sizes the entry point accepts; keeps every
            y1 = y1 < a.src_h - 1 ? y1 : a.src_h - 1;  // load inside the plane regardless
            const int x2 = x1 + 1 < a.src_w - 1 ? x1 + 1 : a.src_w - 1;                                  // :9-10
            const int y2 = y1 + 1 < a.src_h - 1 ? y1 + 1 : a.src_h - 1;
            const float Q11 = (float)a.src[(size_t)y1 * a.src_w + x1], Q21 = (float)a.src[(size_t)y1 * a.src_w + x2];
            const float Q12 = (float)a.src[(size_t)y2 * a.src_w + x1], Q22 = (float)a.src[(size_t)y2 * a.src_w + x2];
            const float dx = (float)(x2 - x1), dy = (float)(y2 - y1);
            const float wx2 = khg_div_f32((float)x2 - x, dx), wx1 = khg_div_f32(x - (float)x1, dx);
            const float wy2 = khg_div_f32((float)y2 - y, dy), wy1 = khg_div_f32(y - (float)y1, dy);
            const float R1 = wx2 * Q11 + wx1 * Q21;  // :17
            const float R2 = wx2 * Q12 + wx1 * Q22;  // :18
            const float P = wy2 * R1 + wy1 * R2;     // :20
            a.dst[p] = khg_f32_to_u8(P);
        }
    }
}
