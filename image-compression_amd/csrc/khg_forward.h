// khg_forward.h -- colour transform + one 2-D CDF 9/7 lifting level + quantiser + histogram.
//
// One workgroup (256 threads) walks down a strip of 256 columns of the plane being
// transformed.  Per batch of 16 source rows:
//   load   : source samples (u8 RGB -> Y / Cb / Cr in f64, or an f64 plane) -> LDS tile A
//   rows   : each thread lifts a run of 16 outputs of one row in registers (window 24)
//            and stores the scaled result de-interleaved (low half | high half) in LDS tile B
//   columns: each thread owns one column, feeds the 16 new rows through a streaming
//            lifting pipeline held in registers, and emits finished coefficients:
//            LL of level 1 goes to an f64 buffer, everything else is quantised to int16
//            symbols and counted in an LDS histogram (count + first raster index).
// Arithmetic order follows DWT.h:7-60 exactly (no FMA contraction: build with
// -ffp-contract=off); see khg_fwd_step.
#pragma once
#include "khg_common.h"

enum { KHG_SRC_Y = 0, KHG_SRC_CB = 1, KHG_SRC_CR = 2, KHG_SRC_PLANE = 3 };

enum {
    KHG_A_COLS = KHG_STRIP + 8,  // strip + 4-sample halo each side
    KHG_A_PITCH = KHG_A_COLS + 1,  // odd pitch: conflict-free ds_read_b64 across 16 rows
    KHG_B_PITCH = KHG_STRIP + 1,
    KHG_LHIST = 1024,            // LDS histogram covers symbols [-512, 511]
    KHG_LHIST_BIAS = 512
};

struct KhgFwdArgs {
    const uint8_t *r, *g, *b;  // full-resolution RGB planes
    int img_w;                 // their row pitch (image columns)
    const double* src;         // KHG_SRC_PLANE: f64 source plane
    int src_pitch;
    int ph, pw;                // rows, cols of this transform
    double* ll;                // level 1: receives the LL quadrant (ph/2 x pw/2)
    int ll_pitch;
    int16_t* sym;              // symbol plane of this colour component
    int sym_pitch;
    double* coef;              // optional f64 tap of every coefficient (may be null)
    int coef_pitch;
    uint32_t* hist_cnt;        // [KHG_NBINS]
    uint32_t* hist_first;      // [KHG_NBINS]
    int* status;
    int lvl;                   // preset 0..2
    int seg_rows;              // rows per workgroup segment (even)
    int nstrip;
    KhgQuant q;
};

struct KhgFwdSmem {
    double A[KHG_BATCH * KHG_A_PITCH];
    double B[KHG_BATCH * KHG_B_PITCH];
    uint32_t cnt[KHG_LHIST];
    uint32_t first[KHG_LHIST];
};

// Streaming state of one 1-D lifting line: values of pair j-1 (e, o, d1, s1) and d2 of pair j-2.
struct KhgLift {
    double e, o, d1, s1, d2;
};

// Feed sample pair j = (x[2j], x[2j+1]); returns the finished, unscaled pair j-2.
// With e[k] = x[2k], o[k] = x[2k+1] the four lifting steps of DWT.h:12-42 read
//   d1[k] = o[k]  + P1*(e[k]    + e[k+1])      predict 1   (DWT.h:13-18)
//   s1[k] = e[k]  + U1*(d1[k-1] + d1[k])       update 1    (DWT.h:21-26)
//   d2[k] = d1[k] + P2*(s1[k]   + s1[k+1])     predict 2   (DWT.h:29-34)
//   s2[k] = s1[k] + U2*(d2[k-1] + d2[k])       update 2    (DWT.h:37-42)
// each evaluated as x + (a * (l + r)), three roundings, like x[i] += a * (x[i-1] + x[i+1]).
KHG_DEV void khg_fwd_step(KhgLift& st, double en, double on, double& s2, double& d2)
{
    const double P1 = -1.586134342, U1 = -0.05298011854, P2 = 0.8829110762, U2 = 0.4435068522;
    double d1n = st.o + P1 * (st.e + en);
    double s1n = st.e + U1 * (st.d1 + d1n);
    double d2n = st.d1 + P2 * (st.s1 + s1n);
    double s2n = st.s1 + U2 * (st.d2 + d2n);
    s2 = s2n;
    d2 = d2n;
    st.e = en;
    st.o = on;
    st.d1 = d1n;
    st.s1 = s1n;
    st.d2 = d2n;
}

// DWT.h:45-50: a = 1/1.149604398 (rounded once); odd samples x*a, even samples x/a.
KHG_DEV double khg_fwd_scale_even(double x) { const double a = 1 / 1.149604398; return x / a; }
KHG_DEV double khg_fwd_scale_odd(double x) { const double a = 1 / 1.149604398; return x * a; }

template <int SRC>
KHG_DEV double khg_fwd_load(const KhgFwdArgs& a, int gy, int gx)
{
    if (SRC == KHG_SRC_PLANE) return a.src[(size_t)gy * a.src_pitch + gx];
    // Main.cpp:200-202 widens u8 to double; Colorspace.h:11,19,20 left-to-right products and sums.
    size_t p = (SRC == KHG_SRC_Y) ? (size_t)gy * a.img_w + gx : (size_t)(2 * gy) * a.img_w + 2 * gx;
    double R = (double)a.r[p], G = (double)a.g[p], B = (double)a.b[p];
    if (SRC == KHG_SRC_Y) return R * 0.299 + G * 0.587 + B * 0.114;
    if (SRC == KHG_SRC_CB) return R * -0.16874 + G * -0.33126 + B * 0.5;
    return R * 0.5 + G * -0.41869 + B * -0.08131;
}

// Per-thread counts and first raster indices of the symbols -1, 0, +1.  These make up almost all
// detail-band symbols; counting them in registers keeps 64 lanes from hammering one LDS word.
struct KhgHot {
    uint32_t cm, c0, cp;
    uint32_t fm, f0, fp;
};

// Quantise one coefficient, store the symbol and count it (Compress.h:218-219, Huffman.h:110-139).
KHG_DEV void khg_quant_store(const KhgFwdArgs& a, KhgFwdSmem& sh, KhgHot& hot, double c, int band, int y, int x)
{
    double div = khg_divisor(a.q, band, x, y, a.lvl);
    int q = (int)round(c / div);
    if (q < -KHG_SYM_BIAS || q >= KHG_SYM_BIAS) {
        khg_set_status(a.status, -4);
        q = 0;
    }
    uint32_t raster = (uint32_t)y * (uint32_t)a.sym_pitch + (uint32_t)x;
    a.sym[raster] = (int16_t)q;
    if ((unsigned)(q + 1) <= 2u) {
        hot.cm += q == -1;
        hot.c0 += q == 0;
        hot.cp += q == 1;
        hot.fm = (q == -1 && raster < hot.fm) ? raster : hot.fm;
        hot.f0 = (q == 0 && raster < hot.f0) ? raster : hot.f0;
        hot.fp = (q == 1 && raster < hot.fp) ? raster : hot.fp;
        return;
    }
    int l = q + KHG_LHIST_BIAS;
    if ((unsigned)l < (unsigned)KHG_LHIST) {
        khg_atomic_add(&sh.cnt[l], 1u);
        khg_atomic_min(&sh.first[l], raster);
    } else {
        khg_atomic_add(&a.hist_cnt[q + KHG_SYM_BIAS], 1u);
        khg_atomic_min(&a.hist_first[q + KHG_SYM_BIAS], raster);
    }
}

// LEVEL 1: bands LH(4) HL(5) HH(6) are final, LL feeds level 2.  LEVEL 2: bands 0..3 (Compress.h:154-160).
template <int SRC, int LEVEL>
KHG_DEV void khg_fwd_strip_body(const KhgFwdArgs& a, int bid, KhgFwdSmem& sh)
{
    const int strip = bid % a.nstrip, seg = bid / a.nstrip;
    const int x0 = strip * KHG_STRIP;
    const int r0 = seg * a.seg_rows;
    const int r1 = (r0 + a.seg_rows < a.ph) ? r0 + a.seg_rows : a.ph;
    const int rin0 = r0 - 4;
    const int k0 = r0 / 2, k1 = r1 / 2;
    const int nb = (r1 - r0 + 8 + KHG_BATCH - 1) / KHG_BATCH;
    const int hh = a.ph / 2, hw = a.pw / 2;

    KhgLift cs[KHG_NSLOT(256)];
    KhgHot hots[KHG_NSLOT(256)];
    KHG_FOR_THREADS(256)
    {
        KhgLift z = {0, 0, 0, 0, 0};
        cs[KHG_SLOT] = z;
        KhgHot hz = {0, 0, 0, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
        hots[KHG_SLOT] = hz;
        for (int i = tid; i < KHG_LHIST; i += 256) {
            sh.cnt[i] = 0;
            sh.first[i] = 0xFFFFFFFFu;
        }
    }
    KHG_BARRIER();

    for (int b = 0; b <= nb; ++b) {
        KHG_FOR_THREADS(256)
        {
            if (b > 0) {
                // ---- column phase of batch b-1: thread tid owns de-interleaved column tid
                const int half = tid >> 7, tt = tid & 127;
                const int c = 2 * tt + half;  // strip-local source column
                const bool colok = x0 + c < a.pw;
                const int xq = (x0 >> 1) + tt;
                const int xo = half ? hw + xq : xq;  // packed output column
                KhgLift st = cs[KHG_SLOT];
                KhgHot hot = hots[KHG_SLOT];
                for (int rr = 0; rr < KHG_BATCH; rr += 2) {
                    double en = sh.B[rr * KHG_B_PITCH + tid];
                    double on = sh.B[(rr + 1) * KHG_B_PITCH + tid];
                    double s2, d2;
                    khg_fwd_step(st, en, on, s2, d2);
                    int k = (rin0 + (b - 1) * KHG_BATCH + rr) / 2 - 2;  // numerator is even
                    if (colok && k >= k0 && k < k1) {
                        double lo = khg_fwd_scale_even(s2), hi = khg_fwd_scale_odd(d2);
                        if (a.coef) {
                            a.coef[(size_t)k * a.coef_pitch + xo] = lo;
                            a.coef[(size_t)(hh + k) * a.coef_pitch + xo] = hi;
                        }
                        if (LEVEL == 1) {
                            if (half == 0) {
                                a.ll[(size_t)k * a.ll_pitch + xq] = lo;
                                khg_quant_store(a, sh, hot, hi, 5, hh + k, xo);
                            } else {
                                khg_quant_store(a, sh, hot, lo, 4, k, xo);
                                khg_quant_store(a, sh, hot, hi, 6, hh + k, xo);
                            }
                        } else {
                            if (half == 0) {
                                khg_quant_store(a, sh, hot, lo, 0, k, xo);
                                khg_quant_store(a, sh, hot, hi, 2, hh + k, xo);
                            } else {
                                khg_quant_store(a, sh, hot, lo, 1, k, xo);
                                khg_quant_store(a, sh, hot, hi, 3, hh + k, xo);
                            }
                        }
                    }
                }
                cs[KHG_SLOT] = st;
                hots[KHG_SLOT] = hot;
            }
            if (b < nb) {
                // ---- load phase of batch b
                for (int e = tid; e < KHG_BATCH * KHG_A_COLS; e += 256) {
                    int rr = e / KHG_A_COLS, i = e - rr * KHG_A_COLS;
                    int gy = khg_reflect(rin0 + b * KHG_BATCH + rr, a.ph);
                    int gx = khg_reflect(x0 - 4 + i, a.pw);
                    sh.A[rr * KHG_A_PITCH + i] = khg_fwd_load<SRC>(a, gy, gx);
                }
            }
        }
        KHG_BARRIER();
        if (b < nb) {
            KHG_FOR_THREADS(256)
            {
                // ---- row phase: row rr, outputs [16*run, 16*run+16) of the strip
                const int rr = tid & 15, run = tid >> 4;
                const double* w = &sh.A[rr * KHG_A_PITCH + run * 16];
                double* blo = &sh.B[rr * KHG_B_PITCH + run * 8];
                double* bhi = blo + 128;
                KhgLift st = {0, 0, 0, 0, 0};
KHG_UNROLL
                for (int j = 0; j < 12; ++j) {
                    double s2, d2;
                    khg_fwd_step(st, w[2 * j], w[2 * j + 1], s2, d2);
                    if (j >= 4) {
                        blo[j - 4] = khg_fwd_scale_even(s2);
                        bhi[j - 4] = khg_fwd_scale_odd(d2);
                    }
                }
            }
            KHG_BARRIER();
        }
    }

    KHG_FOR_THREADS(256)
    {
        const KhgHot hot = hots[KHG_SLOT];
        if (hot.cm) {
            khg_atomic_add(&sh.cnt[KHG_LHIST_BIAS - 1], hot.cm);
            khg_atomic_min(&sh.first[KHG_LHIST_BIAS - 1], hot.fm);
        }
        if (hot.c0) {
            khg_atomic_add(&sh.cnt[KHG_LHIST_BIAS], hot.c0);
            khg_atomic_min(&sh.first[KHG_LHIST_BIAS], hot.f0);
        }
        if (hot.cp) {
            khg_atomic_add(&sh.cnt[KHG_LHIST_BIAS + 1], hot.cp);
            khg_atomic_min(&sh.first[KHG_LHIST_BIAS + 1], hot.fp);
        }
    }
    KHG_BARRIER();
    KHG_FOR_THREADS(256)
    {
        for (int i = tid; i < KHG_LHIST; i += 256) {
            uint32_t c = sh.cnt[i];
            if (c) {
                int bin = i - KHG_LHIST_BIAS + KHG_SYM_BIAS;
                khg_atomic_add(&a.hist_cnt[bin], c);
                khg_atomic_min(&a.hist_first[bin], sh.first[i]);
            }
        }
    }
}
