// khg_inverse.h -- dequantiser + one inverse 2-D CDF 9/7 lifting level, and the final
// YCbCr -> RGB conversion.
//
// Mirror image of khg_forward.h (IWT2D runs columns first, then rows: DWT.h:143-165).
// One workgroup walks down a strip of 240 output columns.  Per batch of 8 row pairs:
//   columns: 248 threads each own one packed source column (124 low + 124 high, i.e. the
//            strip's 120 + 120 plus a 2-pair halo each side); a thread reads its low/high
//            coefficient pair (symbols are dequantised on the fly, Decompress.h:98-123),
//            feeds the streaming inverse lifting pipeline in registers, and stores the two
//            finished natural-order rows in LDS tile T
//   rows   : each thread inverse-lifts a run of 16 outputs of one row from 12 low + 12 high
//            values of T and writes them to the destination plane -- or, for the luma plane at level 1,
//            converts them with the finished chroma planes to RGB (Colorspace.h:25-36) and writes the
//            u8 image directly, so the f64 luma plane never exists in memory.
#pragma once
#include "khg_common.h"
#include "khg_decode.h"

enum {
    KHG_ISTRIP = 240,
    KHG_IHALF = KHG_ISTRIP / 2 + 4,   // 124 packed columns per half (2-pair halo each side)
    KHG_ICOLS = 2 * KHG_IHALF,        // 248 column threads
    KHG_T_PITCH = KHG_ICOLS + 1,
    KHG_IBATCH = 8                    // row pairs per batch
};

struct KhgInvArgs {
    const int16_t* sym;   // symbol plane of this colour component
    int sym_pitch;
    const double* ll;     // level 1: the LL quadrant (output of the level-2 inverse)
    int ll_pitch;
    int ph, pw;           // rows, cols of this transform
    double* dst;          // natural-order output, ph x pw
    int dst_pitch;
    const KhgDecHeader* hdr;  // supplies the preset (lvl) and the go/no-go flag
    int row_begin, row_end;  // output rows this launch produces (even; the whole plane, or one tile's band)
    int out_row0;         // RGB variant: image row held by the first row of r/g/b (0 for a whole image)
    int seg_rows;         // output rows per workgroup segment (even)
    int nstrip;
    KhgQuant q[3];        // quantiser constants of this plane type for presets 0..2
    // RGB variant (level 1 of the luma plane): the finished chroma planes and the output image
    const double *cb, *cr;  // (ph/2) x (pw/2)
    uint8_t *r, *g, *b;     // ph x pw, pitch pw
};

struct KhgInvSmem {
    double T[2 * KHG_IBATCH * KHG_T_PITCH];
};

// State after feeding pair j: d_in[j], s1[j], d1[j-1], s0[j-1].
struct KhgILift {
    double d, s1, d1, s0;
};

// Feed packed pair j = (low[j], high[j]); returns natural samples x[2(j-2)], x[2(j-2)+1].
// With s[k] = low[k]/A, d[k] = high[k]*A (DWT.h:76-82, A = 1.149604398) the steps of DWT.h:84-114 read
//   s1[k] = s[k]  + C1*(d[k-1]  + d[k])        undo update 2
//   d1[k] = d[k]  + C2*(s1[k]   + s1[k+1])     undo predict 2
//   s0[k] = s1[k] + C3*(d1[k-1] + d1[k])       undo update 1
//   d0[k] = d1[k] + C4*(s0[k]   + s0[k+1])     undo predict 1
KHG_DEV void khg_inv_step(KhgILift& st, double lo, double hi, double& xe, double& xo)
{
    const double A = 1.149604398;
    const double C1 = -0.4435068522, C2 = -0.8829110762, C3 = 0.05298011854, C4 = 1.586134342;
    const double rA = 1 / A;
    double sn = khg_div_const(lo, A, rA), dn = hi * A;
    double s1n = sn + C1 * (st.d + dn);
    double d1n = st.d + C2 * (st.s1 + s1n);
    double s0n = st.s1 + C3 * (st.d1 + d1n);
    double d0 = st.d1 + C4 * (st.s0 + s0n);
    xe = st.s0;
    xo = d0;
    st.d = dn;
    st.s1 = s1n;
    st.d1 = d1n;
    st.s0 = s0n;
}

// Raw inputs of one thread's column for one batch: symbols as stored, LL values as f64.  Fetched one batch
// ahead (khg_inv_fetch) so the loads fly during the row phase, turned into coefficients when the column phase
// consumes them (khg_inv_coef).
struct KhgInvRaw {
    int q[2 * KHG_IBATCH];     // [2s] low-pass row of pair s, [2s+1] its high-pass row
    double ll[KHG_IBATCH];     // level 1, low-pass row of a left-half column: the LL sample instead of a symbol
};

// Band of packed position (yo, xo), -1 for the level-1 LL quadrant.
template <int LEVEL>
KHG_DEV int khg_inv_band(int yo, int xo, int hh, int hw)
{
    if (LEVEL == 2) return (yo >= hh ? 2 : 0) + (xo >= hw ? 1 : 0);
    if (yo < hh && xo < hw) return -1;
    return yo < hh ? 4 : (xo < hw ? 5 : 6);
}

template <int LEVEL>
KHG_DEV void khg_inv_fetch(const KhgInvArgs& a, KhgInvRaw& raw, int s, int ye, int yo, int xo, int hh, int hw)
{
    if (khg_inv_band<LEVEL>(ye, xo, hh, hw) < 0) {
        raw.ll[s] = a.ll[(size_t)ye * a.ll_pitch + xo];
        raw.q[2 * s] = 0;
    } else {
        raw.ll[s] = 0.0;
        raw.q[2 * s] = a.sym[(size_t)ye * a.sym_pitch + xo];
    }
    raw.q[2 * s + 1] = a.sym[(size_t)yo * a.sym_pitch + xo];
}

// Decompress.h:105, 120-121: coefficient = symbol * divisor of its band and position.
template <int LEVEL>
KHG_DEV double khg_inv_coef(const KhgQuant& qc, int lvl, int q, double ll, int y, int xo, int hh, int hw)
{
    const int band = khg_inv_band<LEVEL>(y, xo, hh, hw);
    if (band < 0) return ll;
    return (double)q * khg_divisor(qc, band, xo, y, lvl);
}

KHG_DEV uint32_t khg_clamp_u8(double v)
{
    // Main.cpp:307-315: max(0, min(v, 255)) with the Win32 macros (a<b?a:b), (a>b?a:b), then truncation to
    // unsigned char.  v is never NaN here (finite symbols times finite constants), and for every other v
    // fmin/fmax pick the same operand as the macros, up to the sign of a zero that the truncation drops.
    return (uint32_t)fmax(0.0, fmin(v, 255.0));
}

// Colorspace.h:25-36 + Main.cpp:307-315 for the two pixels that share one chroma site: products and sums
// left to right, clamp, truncate.  Returns the pixels' bytes in bits 0..7 and 8..15 of rgb[0..2].
KHG_DEV void khg_rgb_pair(double y0, double y1, double Cb, double Cr, uint32_t rgb[3])
{
    const double rr = Cr * 1.402, gb = Cb * -0.34414, gr = Cr * -0.71414, bb = Cb * 1.772;
    rgb[0] = (uint32_t)khg_clamp_u8(y0 * 1 + rr) | ((uint32_t)khg_clamp_u8(y1 * 1 + rr) << 8);
    rgb[1] = (uint32_t)khg_clamp_u8(y0 * 1 + gb + gr) | ((uint32_t)khg_clamp_u8(y1 * 1 + gb + gr) << 8);
    rgb[2] = (uint32_t)khg_clamp_u8(y0 * 1 + bb) | ((uint32_t)khg_clamp_u8(y1 * 1 + bb) << 8);
}

// RGB = true (level 1 of the luma plane only): the row phase does not store the luma samples but combines
// them with the already reconstructed chroma planes and writes the u8 image.
template <int LEVEL, bool RGB = false>
KHG_DEV void khg_inv_strip_body(const KhgInvArgs& a, int bid, KhgInvSmem& sh)
{
    const int strip = bid % a.nstrip, seg = bid / a.nstrip;
    const int x0 = strip * KHG_ISTRIP;
    const int r0 = a.row_begin + seg * a.seg_rows;
    const int r1 = (r0 + a.seg_rows < a.row_end) ? r0 + a.seg_rows : a.row_end;
    const int k0 = r0 / 2, k1 = r1 / 2;
    const int nb = (k1 - k0 + 4 + KHG_IBATCH - 1) / KHG_IBATCH;
    const int hh = a.ph / 2, hw = a.pw / 2;
    if (!a.hdr->ok) return;
    const int lvl = a.hdr->lvl;
    const KhgQuant& qc = a.q[lvl];

    // The RGB variant's row phase is register-hungry; holding a prefetched batch on top would cost it a wave
    // per SIMD, which hurts more than the prefetch helps, so it loads at the start of the column phase.
    const bool PREFETCH = !RGB;
    KhgILift cs[KHG_NSLOT(256)];
    KhgInvRaw raws[KHG_NSLOT(256)];
    uint32_t touch[KHG_NSLOT(256)] = {0};  // see KHG_TOUCH
    KHG_FOR_THREADS(256)
    {
        KhgILift z = {0, 0, 0, 0};
        cs[KHG_SLOT] = z;
        if (PREFETCH && tid < KHG_ICOLS) {  // inputs of the first batch
            const int half = tid >= KHG_IHALF, tt = half ? tid - KHG_IHALF : tid;
            const int rx = khg_reflect(2 * ((x0 >> 1) - 2 + tt) + half, a.pw);
            const int xo = (rx >> 1) + (half ? hw : 0);
            KHG_UNROLL
            for (int s = 0; s < KHG_IBATCH; ++s) {
                int j = k0 - 2 + s;
                khg_inv_fetch<LEVEL>(a, raws[KHG_SLOT], s, khg_reflect(2 * j, a.ph) >> 1,
                                     hh + (khg_reflect(2 * j + 1, a.ph) >> 1), xo, hh, hw);
            }
        }
    }
    for (int b = 0; b < nb; ++b) {
        KHG_FOR_THREADS(256)
        {
            if (tid < KHG_ICOLS) {
                // ---- column phase: packed source column of this thread
                const int half = tid >= KHG_IHALF, tt = half ? tid - KHG_IHALF : tid;
                const int rx = khg_reflect(2 * ((x0 >> 1) - 2 + tt) + half, a.pw);  // natural column, parity = half
                const int xo = (rx >> 1) + (half ? hw : 0);
                KhgILift st = cs[KHG_SLOT];
                KhgInvRaw& raw = raws[KHG_SLOT];
                if (!PREFETCH) {  // all loads of the batch first (independent), then the lifting chain
                    KHG_UNROLL
                    for (int s = 0; s < KHG_IBATCH; ++s) {
                        int j = k0 - 2 + b * KHG_IBATCH + s;
                        khg_inv_fetch<LEVEL>(a, raw, s, khg_reflect(2 * j, a.ph) >> 1,
                                             hh + (khg_reflect(2 * j + 1, a.ph) >> 1), xo, hh, hw);
                    }
                }
                KHG_UNROLL
                for (int s = 0; s < KHG_IBATCH; ++s) {
                    int j = k0 - 2 + b * KHG_IBATCH + s;  // row pair fed in this step
                    int ye = khg_reflect(2 * j, a.ph) >> 1;
                    int yo = hh + (khg_reflect(2 * j + 1, a.ph) >> 1);
                    double lo = khg_inv_coef<LEVEL>(qc, lvl, raw.q[2 * s], raw.ll[s], ye, xo, hh, hw);
                    double hi = khg_inv_coef<LEVEL>(qc, lvl, raw.q[2 * s + 1], 0.0, yo, xo, hh, hw);
                    double xe, xod;
                    khg_inv_step(st, lo, hi, xe, xod);
                    sh.T[(2 * s) * KHG_T_PITCH + tid] = xe;
                    sh.T[(2 * s + 1) * KHG_T_PITCH + tid] = xod;
                }
                cs[KHG_SLOT] = st;
                if (!PREFETCH && b + 1 < nb) {
                    // No registers to hold the next batch in: at least bring its lines closer.  A wave's 64
                    // columns span two 128-byte lines of a symbol row and five 64-byte pieces of an LL row; lane l
                    // touches row slot l % 16 (symbols; first and last 16 lanes) or LL row l % 8 (every lane), each
                    // at its own column.
                    const int l = tid & 63;
                    const int jn = k0 - 2 + (b + 1) * KHG_IBATCH;
                    const bool llrows = khg_inv_band<LEVEL>(0, xo, hh, hw) < 0;  // low-pass rows come from the LL plane
                    if (llrows) {
                        const int ye = khg_reflect(2 * (jn + (l & 7)), a.ph) >> 1;
                        KHG_TOUCH(touch[KHG_SLOT], a.ll + (size_t)ye * a.ll_pitch + xo);
                    }
                    if (l < 16 || l >= 48) {
                        const int slot = l & 15, j = jn + (slot >> 1);
                        const int row = (slot & 1) ? hh + (khg_reflect(2 * j + 1, a.ph) >> 1) : khg_reflect(2 * j, a.ph) >> 1;
                        if ((slot & 1) || !llrows) KHG_TOUCH(touch[KHG_SLOT], a.sym + (size_t)row * a.sym_pitch + xo);
                    }
                }
                if (PREFETCH && b + 1 < nb) {  // next batch's inputs: in flight during the row phase
                    KHG_UNROLL
                    for (int s = 0; s < KHG_IBATCH; ++s) {
                        int j = k0 - 2 + (b + 1) * KHG_IBATCH + s;
                        khg_inv_fetch<LEVEL>(a, raw, s, khg_reflect(2 * j, a.ph) >> 1,
                                             hh + (khg_reflect(2 * j + 1, a.ph) >> 1), xo, hh, hw);
                    }
                }
            }
        }
        KHG_BARRIER();
        KHG_FOR_THREADS(256)
        {
            // ---- row phase: natural row of T, outputs [16*run, 16*run+16) of the strip
            const int rr = tid & 15, run = tid >> 4;
            const int pair = k0 - 4 + b * KHG_IBATCH + (rr >> 1);  // column phase step s emitted pair j-2
            const int gy = 2 * pair + (rr & 1);
            const int gx0 = x0 + run * 16;
            if (run < KHG_ISTRIP / 16 && pair >= k0 && pair < k1 && gx0 < a.pw) {
                const double* lo = &sh.T[rr * KHG_T_PITCH + run * 8];
                const double* hi = lo + KHG_IHALF;
                double* out = RGB ? nullptr : a.dst + (size_t)gy * a.dst_pitch + gx0;
                const int nv = a.pw - gx0 >= 16 ? 16 : a.pw - gx0;  // 8 or 16 outputs are inside (pw = 0 mod 8)
                double cbv[8], crv[8], yv[16];
                if (RGB) {  // issued first: in flight during the lifting below
                    const size_t cpos = (size_t)(gy >> 1) * (a.pw >> 1) + (gx0 >> 1);
                    if (nv == 16) {  // 64 contiguous, 64-byte aligned bytes per plane: four 16-byte loads each
                        __builtin_memcpy(cbv, a.cb + cpos, 64);
                        __builtin_memcpy(crv, a.cr + cpos, 64);
                    } else {
                        KHG_UNROLL
                        for (int c = 0; c < 8; ++c) {
                            cbv[c] = 2 * c < nv ? a.cb[cpos + c] : 0.0;
                            crv[c] = 2 * c < nv ? a.cr[cpos + c] : 0.0;
                        }
                    }
                }
                KhgILift st = {0, 0, 0, 0};
                KHG_UNROLL
                for (int i = 0; i < 12; ++i) {
                    double xe, xod;
                    khg_inv_step(st, lo[i], hi[i], xe, xod);
                    if (RGB) {
                        if (i >= 4) {
                            yv[2 * (i - 4)] = xe;
                            yv[2 * (i - 4) + 1] = xod;
                        }
                    } else if (i >= 4 && 2 * (i - 4) < nv) {  // pw is even: a pair never straddles the edge
                        out[2 * (i - 4)] = xe;
                        out[2 * (i - 4) + 1] = xod;
                    }
                }
                if (RGB) {
                    uint64_t pk[3][2] = {{0, 0}, {0, 0}, {0, 0}};  // 8 pixels per 64-bit word, first pixel lowest
                    KHG_UNROLL
                    for (int c = 0; c < 8; ++c) {
                        uint32_t rgb[3];
                        khg_rgb_pair(yv[2 * c], yv[2 * c + 1], cbv[c], crv[c], rgb);
                        for (int ch = 0; ch < 3; ++ch) pk[ch][c >> 2] |= (uint64_t)rgb[ch] << (16 * (c & 3));
                    }
                    const size_t p = (size_t)(gy - a.out_row0) * a.pw + gx0;
                    if (nv == 16) {  // one 16-byte store per plane
                        __builtin_memcpy(a.r + p, pk[0], 16);
                        __builtin_memcpy(a.g + p, pk[1], 16);
                        __builtin_memcpy(a.b + p, pk[2], 16);
                    } else {
                        __builtin_memcpy(a.r + p, pk[0], 8);
                        __builtin_memcpy(a.g + p, pk[1], 8);
                        __builtin_memcpy(a.b + p, pk[2], 8);
                    }
                }
            }
        }
        KHG_BARRIER();
    }
    if (!PREFETCH) {
        KHG_FOR_THREADS(256) { KHG_TOUCH_DONE(touch[KHG_SLOT]); }
    }
}

