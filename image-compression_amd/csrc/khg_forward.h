// khg_forward.h -- colour transform + one 2-D CDF 9/7 lifting level + quantiser + histogram.
//
// One workgroup (256 threads) walks down a strip of 256 columns of the plane being
// transformed.  Per batch of 16 source rows:
//   rows   : each thread loads the 24-sample window of a run of 16 outputs of one row straight into
//            registers (u8 RGB as packed words, converted to Y / Cb / Cr in f64, or an f64 plane), one
//            batch ahead so the loads fly during the column phase, lifts it and stores the scaled
//            result de-interleaved (low half | high half) in LDS tile B
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
    KHG_B_PITCH = KHG_STRIP + 1,  // odd pitch: conflict-free ds_write_b64 across the 16 rows of a wave
    KHG_MAX_SEG_ROWS = 128,      // rows per workgroup segment, at most
    KHG_RD_N = KHG_STRIP / 2 + KHG_MAX_SEG_ROWS / 2  // reciprocal-divisor entries per band
};

struct KhgFwdArgs {
    const uint8_t *r, *g, *b;  // full-resolution RGB planes, or a band of their rows
    int img_w;                 // their row pitch (image columns)
    int in_row0;               // image row held by the first row of r/g/b (0 for a whole image)
    int in_rows;               // image rows r/g/b hold
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
    KhgResult* prof;           // when set, the launch's first start / last end device clock go here
    int lvl;                   // preset 0..2
    int row_begin, row_end;    // plane rows this launch transforms (even; the whole plane, or one tile's band)
    int emit_k0, emit_k1;      // row pairs whose detail symbols are stored and counted (a tile recomputes its
                               // neighbours' edge rows for the LL quadrant only)
    int seg_rows;              // rows per workgroup segment (even)
    int nstrip;
    KhgQuant q;
};

// LDS of one workgroup.  Level 1 emits detail bands only (small symbols, three bands): a 256-bin histogram and
// three reciprocal tables keep it under 40 KB, i.e. four workgroups per CU.  Level 2 also emits the LL band,
// whose symbols spread over hundreds of values: 1024 bins, four tables.
template <int LEVEL>
struct KhgFwdSmem {
    enum { LHIST = LEVEL == 1 ? 256 : 1024, LHIST_BIAS = LHIST / 2, NBAND = LEVEL == 1 ? 3 : 4 };
    double B[KHG_BATCH * KHG_B_PITCH];
    uint32_t cnt[LHIST];    // symbols [-LHIST/2, LHIST/2)
    uint32_t first[LHIST];
    double rdiv[NBAND * KHG_RD_N];  // 1/divisor of band slot s at diagonal u: [s * KHG_RD_N + u]
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
KHG_DEV double khg_fwd_scale_even(double x)
{
    const double a = 1 / 1.149604398, ra = 1 / a;
    return khg_div_const(x, a, ra);
}
KHG_DEV double khg_fwd_scale_odd(double x) { const double a = 1 / 1.149604398; return x * a; }

// ---- source window of one row-phase thread: 24 consecutive samples of one row, as loaded.
// u8 RGB sources keep the raw bytes (4 luma samples per 32-bit word; 4 chroma sites per 64-bit word, every
// second byte) so the loads are issued a batch ahead and converted when the row phase needs them.
template <int SRC>
struct KhgFwdRaw {
    uint32_t w[SRC == KHG_SRC_Y ? 18 : SRC == KHG_SRC_PLANE ? 48 : 36];
};

KHG_DEV uint32_t khg_ld32(const uint8_t* p) { return *(const uint32_t*)p; }

// Image row `row` as a row of the r/g/b buffers.  The last batch of a segment runs past the rows its outputs need;
// for a band of rows (a tile) those surplus rows may lie outside the band, so they are clamped into it -- their
// results are discarded either way.
KHG_DEV int khg_band_row(const KhgFwdArgs& a, int row)
{
    int r = row - a.in_row0;
    return r < 0 ? 0 : (r >= a.in_rows ? a.in_rows - 1 : r);
}

// Row gy of the plane, samples gx0 .. gx0+23 (gx0 = 0 mod 4; columns outside [0, pw) are reflected).
template <int SRC>
KHG_DEV void khg_fwd_fetch(const KhgFwdArgs& a, KhgFwdRaw<SRC>& raw, int gy, int gx0)
{
    if (SRC == KHG_SRC_PLANE) {
        const double* row = a.src + (size_t)gy * a.src_pitch;
        KHG_UNROLL
        for (int k = 0; k < 12; ++k) {
            int gx = gx0 + 2 * k;
            double v0, v1;
            if (gx >= 0 && gx + 1 < a.pw) {  // pw is even: a pair is inside or outside as a whole
                double two[2];
                __builtin_memcpy(two, row + gx, 16);  // one 16-byte load
                v0 = two[0];
                v1 = two[1];
            } else {
                v0 = row[khg_reflect(gx, a.pw)];
                v1 = row[khg_reflect(gx + 1, a.pw)];
            }
            __builtin_memcpy(&raw.w[4 * k], &v0, 8);
            __builtin_memcpy(&raw.w[4 * k + 2], &v1, 8);
        }
    } else if (SRC == KHG_SRC_Y) {
        const size_t rowoff = (size_t)khg_band_row(a, gy) * a.img_w;
        KHG_UNROLL
        for (int k = 0; k < 6; ++k) {
            int gx = gx0 + 4 * k;
            if (gx >= 0 && gx + 3 < a.pw) {  // pw = 0 mod 8
                raw.w[k] = khg_ld32(a.r + rowoff + gx);
                raw.w[6 + k] = khg_ld32(a.g + rowoff + gx);
                raw.w[12 + k] = khg_ld32(a.b + rowoff + gx);
            } else {
                uint32_t r = 0, g = 0, b = 0;
                for (int i = 0; i < 4; ++i) {
                    size_t p = rowoff + khg_reflect(gx + i, a.pw);
                    r |= (uint32_t)a.r[p] << (8 * i);
                    g |= (uint32_t)a.g[p] << (8 * i);
                    b |= (uint32_t)a.b[p] << (8 * i);
                }
                raw.w[k] = r; raw.w[6 + k] = g; raw.w[12 + k] = b;
            }
        }
    } else {
        // chroma site (gy, gx) reads the pixel at (2gy, 2gx): Colorspace.h:16-21
        const size_t rowoff = (size_t)khg_band_row(a, 2 * gy) * a.img_w;
        KHG_UNROLL
        for (int k = 0; k < 6; ++k) {
            int gx = gx0 + 4 * k;
            if (gx >= 0 && gx + 3 < a.pw) {  // pw = 0 mod 4
                size_t p = rowoff + 2 * gx;
                raw.w[2 * k] = khg_ld32(a.r + p);       raw.w[2 * k + 1] = khg_ld32(a.r + p + 4);
                raw.w[12 + 2 * k] = khg_ld32(a.g + p);  raw.w[12 + 2 * k + 1] = khg_ld32(a.g + p + 4);
                raw.w[24 + 2 * k] = khg_ld32(a.b + p);  raw.w[24 + 2 * k + 1] = khg_ld32(a.b + p + 4);
            } else {
                uint32_t r[2] = {0, 0}, g[2] = {0, 0}, b[2] = {0, 0};
                for (int i = 0; i < 4; ++i) {
                    size_t p = rowoff + 2 * khg_reflect(gx + i, a.pw);
                    r[i >> 1] |= (uint32_t)a.r[p] << (16 * (i & 1));
                    g[i >> 1] |= (uint32_t)a.g[p] << (16 * (i & 1));
                    b[i >> 1] |= (uint32_t)a.b[p] << (16 * (i & 1));
                }
                raw.w[2 * k] = r[0]; raw.w[2 * k + 1] = r[1];
                raw.w[12 + 2 * k] = g[0]; raw.w[12 + 2 * k + 1] = g[1];
                raw.w[24 + 2 * k] = b[0]; raw.w[24 + 2 * k + 1] = b[1];
            }
        }
    }
}

// Sample i (0..23) of a fetched window as the f64 value the transform runs on.
// Main.cpp:200-202 widens u8 to double; Colorspace.h:11,19,20 left-to-right products and sums.
template <int SRC>
KHG_DEV double khg_fwd_sample(const KhgFwdRaw<SRC>& raw, int i)
{
    if (SRC == KHG_SRC_PLANE) {
        double v;
        __builtin_memcpy(&v, &raw.w[2 * i], 8);
        return v;
    }
    double R, G, B;
    if (SRC == KHG_SRC_Y) {
        int k = i >> 2, sh = 8 * (i & 3);
        R = (double)((raw.w[k] >> sh) & 255u);
        G = (double)((raw.w[6 + k] >> sh) & 255u);
        B = (double)((raw.w[12 + k] >> sh) & 255u);
        return R * 0.299 + G * 0.587 + B * 0.114;
    }
    int k = i >> 1, sh = 16 * (i & 1);
    R = (double)((raw.w[k] >> sh) & 255u);
    G = (double)((raw.w[12 + k] >> sh) & 255u);
    B = (double)((raw.w[24 + k] >> sh) & 255u);
    if (SRC == KHG_SRC_CB) return R * -0.16874 + G * -0.33126 + B * 0.5;
    return R * 0.5 + G * -0.41869 + B * -0.08131;
}

// Per-thread tally of the symbols -1, 0, +1, which make up almost all detail-band symbols: counting them
// in registers keeps 64 lanes from hammering one LDS word.  cnt packs three 21-bit counters (a thread
// emits far fewer symbols per segment); seen has one bit per (symbol, output stream).  A thread's raster
// indices rise within each of its two output streams (rows k and rows hh+k), so only the first hit of a
// symbol in a stream can lower the workgroup's first-appearance index: one LDS atomic per bit, ever.
struct KhgHot {
    uint64_t cnt;
    uint32_t seen;
};

// Quantise one coefficient, store the symbol and count it (Compress.h:218-219, Huffman.h:110-139).
// rdiv = RN(1/divisor) of this position (khg_round_quot_fast); stream 0 = rows k, 1 = rows hh+k.
template <class SMEM>
KHG_DEV void khg_quant_store(const KhgFwdArgs& a, SMEM& sh, KhgHot& hot, double c, double rdiv, int band,
                             int stream, int y, int x)
{
    double n;
    if (!khg_round_quot_fast(c, rdiv, &n)) n = round(c / khg_divisor(a.q, band, x, y, a.lvl));
    if (!(n >= -(double)KHG_SYM_BIAS && n < (double)KHG_SYM_BIAS)) {
        khg_set_status(a.status, -4);
        n = 0;
    }
    int q = (int)n;
    uint32_t raster = (uint32_t)y * (uint32_t)a.sym_pitch + (uint32_t)x;
    a.sym[raster] = (int16_t)q;
    int l = q + SMEM::LHIST_BIAS;
    if ((unsigned)(q + 1) <= 2u) {
        hot.cnt += 1ull << (21 * (q + 1));
        uint32_t bit = 1u << (2 * (q + 1) + stream);
        if (!(hot.seen & bit)) {
            hot.seen |= bit;
            khg_atomic_min(&sh.first[l], raster);
        }
    } else if ((unsigned)l < (unsigned)SMEM::LHIST) {
        khg_atomic_add(&sh.cnt[l], 1u);
        khg_atomic_min(&sh.first[l], raster);
    } else {
        khg_atomic_add(&a.hist_cnt[q + KHG_SYM_BIAS], 1u);
        khg_atomic_min(&a.hist_first[q + KHG_SYM_BIAS], raster);
    }
}

// LEVEL 1: bands LH(4) HL(5) HH(6) are final, LL feeds level 2.  LEVEL 2: bands 0..3 (Compress.h:154-160).
template <int SRC, int LEVEL>
KHG_DEV void khg_fwd_strip_body(const KhgFwdArgs& a, int bid, KhgFwdSmem<LEVEL>& sh)
{
    const int strip = bid % a.nstrip, seg = bid / a.nstrip;
    const int x0 = strip * KHG_STRIP;
    const int r0 = a.row_begin + seg * a.seg_rows;
    const int r1 = (r0 + a.seg_rows < a.row_end) ? r0 + a.seg_rows : a.row_end;
    const int rin0 = r0 - 4;
    const int k0 = r0 / 2, k1 = r1 / 2;
    const int nb = (r1 - r0 + 8 + KHG_BATCH - 1) / KHG_BATCH;
    const int hh = a.ph / 2, hw = a.pw / 2;

    KhgLift cs[KHG_NSLOT(256)];
    KhgHot hots[KHG_NSLOT(256)];
    KhgFwdRaw<SRC> raws[KHG_NSLOT(256)];
    KHG_FOR_THREADS(256)
    {
        if (a.prof && tid == 0) khg_atomic_max64(&a.prof->prof_not_start, ~khg_clock());
        KhgLift z = {0, 0, 0, 0, 0};
        cs[KHG_SLOT] = z;
        KhgHot hz = {0, 0};
        hots[KHG_SLOT] = hz;
        // window of the first batch (row-phase mapping: row tid & 15, run tid >> 4)
        khg_fwd_fetch<SRC>(a, raws[KHG_SLOT], khg_reflect(rin0 + (tid & 15), a.ph), x0 + (tid >> 4) * 16 - 4);
        for (int i = tid; i < KhgFwdSmem<LEVEL>::LHIST; i += 256) {
            sh.cnt[i] = 0;
            sh.first[i] = 0xFFFFFFFFu;
        }
        // The divisor depends on the band and on x + y only (Compress.h:215-218); column c of the strip at
        // row pair k sits on diagonal u = c/2 + (k - k0) of each band this workgroup emits.
        const int nu = KHG_STRIP / 2 + (k1 - k0);
        for (int e = tid; e < KhgFwdSmem<LEVEL>::NBAND * nu; e += 256) {
            int s = e / nu, u = e - s * nu;
            int band = LEVEL == 1 ? 4 + s : s;
            int by = (LEVEL == 1 ? (band >= 5) : (band >= 2)) ? hh : 0;
            int bx = (LEVEL == 1 ? (band != 5) : (band & 1)) ? hw : 0;
            sh.rdiv[s * KHG_RD_N + u] = 1.0 / khg_divisor(a.q, band, bx + (x0 >> 1) + u, by + k0, a.lvl);
        }
    }
    KHG_BARRIER();

    for (int b = 0; b < nb; ++b) {
        KHG_FOR_THREADS(256)
        {
            // ---- row phase: row rr of the batch, outputs [16*run, 16*run+16) of the strip
            const int rr = tid & 15, run = tid >> 4;
            KhgFwdRaw<SRC>& raw = raws[KHG_SLOT];
            double* blo = &sh.B[rr * KHG_B_PITCH + run * 8];
            double* bhi = blo + 128;
            KhgLift st = {0, 0, 0, 0, 0};
            KHG_UNROLL
            for (int j = 0; j < 12; ++j) {
                double s2, d2;
                khg_fwd_step(st, khg_fwd_sample<SRC>(raw, 2 * j), khg_fwd_sample<SRC>(raw, 2 * j + 1), s2, d2);
                if (j >= 4) {
                    blo[j - 4] = khg_fwd_scale_even(s2);
                    bhi[j - 4] = khg_fwd_scale_odd(d2);
                }
            }
            // next batch's window: in flight while the column phase runs
            if (b + 1 < nb)
                khg_fwd_fetch<SRC>(a, raw, khg_reflect(rin0 + (b + 1) * KHG_BATCH + rr, a.ph), x0 + run * 16 - 4);
        }
        KHG_BARRIER();
        KHG_FOR_THREADS(256)
        {
            // ---- column phase: thread tid owns de-interleaved column tid
            const int half = tid >> 7, tt = tid & 127;
            const int c = 2 * tt + half;  // strip-local source column
            const bool colok = x0 + c < a.pw;
            const int xq = (x0 >> 1) + tt;
            const int xo = half ? hw + xq : xq;  // packed output column
            KhgLift st = cs[KHG_SLOT];
            KhgHot hot = hots[KHG_SLOT];
            // kept as a loop: unrolled eight times, the rarely taken division paths of the quantiser alone made
            // the kernel larger than the instruction cache likes
            KHG_NOUNROLL
            for (int rr = 0; rr < KHG_BATCH; rr += 2) {
                double en = sh.B[rr * KHG_B_PITCH + tid];
                double on = sh.B[(rr + 1) * KHG_B_PITCH + tid];
                double s2, d2;
                khg_fwd_step(st, en, on, s2, d2);
                int k = (rin0 + b * KHG_BATCH + rr) / 2 - 2;  // numerator is even
                if (colok && k >= k0 && k < k1) {
                    const double* rd = &sh.rdiv[tt + (k - k0)];
                    double lo = khg_fwd_scale_even(s2), hi = khg_fwd_scale_odd(d2);
                    if (a.coef) {
                        a.coef[(size_t)k * a.coef_pitch + xo] = lo;
                        a.coef[(size_t)(hh + k) * a.coef_pitch + xo] = hi;
                    }
                    const bool emit = k >= a.emit_k0 && k < a.emit_k1;
                    if (LEVEL == 1) {
                        if (half == 0) {
                            a.ll[(size_t)k * a.ll_pitch + xq] = lo;
                            if (emit) khg_quant_store(a, sh, hot, hi, rd[1 * KHG_RD_N], 5, 1, hh + k, xo);
                        } else if (emit) {
                            khg_quant_store(a, sh, hot, lo, rd[0 * KHG_RD_N], 4, 0, k, xo);
                            khg_quant_store(a, sh, hot, hi, rd[2 * KHG_RD_N], 6, 1, hh + k, xo);
                        }
                    } else if (emit) {
                        if (half == 0) {
                            khg_quant_store(a, sh, hot, lo, rd[0 * KHG_RD_N], 0, 0, k, xo);
                            khg_quant_store(a, sh, hot, hi, rd[2 * KHG_RD_N], 2, 1, hh + k, xo);
                        } else {
                            khg_quant_store(a, sh, hot, lo, rd[1 * KHG_RD_N], 1, 0, k, xo);
                            khg_quant_store(a, sh, hot, hi, rd[3 * KHG_RD_N], 3, 1, hh + k, xo);
                        }
                    }
                }
            }
            cs[KHG_SLOT] = st;
            hots[KHG_SLOT] = hot;
        }
        KHG_BARRIER();
    }

    KHG_FOR_THREADS(256)
    {
        const KhgHot hot = hots[KHG_SLOT];
        for (int v = 0; v < 3; ++v) {
            uint32_t c = (uint32_t)(hot.cnt >> (21 * v)) & 0x1FFFFFu;
            if (c) khg_atomic_add(&sh.cnt[KhgFwdSmem<LEVEL>::LHIST_BIAS - 1 + v], c);
        }
    }
    KHG_BARRIER();
    KHG_FOR_THREADS(256)
    {
        for (int i = tid; i < KhgFwdSmem<LEVEL>::LHIST; i += 256) {
            uint32_t c = sh.cnt[i];
            if (c) {
                int bin = i - KhgFwdSmem<LEVEL>::LHIST_BIAS + KHG_SYM_BIAS;
                khg_atomic_add(&a.hist_cnt[bin], c);
                khg_atomic_min(&a.hist_first[bin], sh.first[i]);
            }
        }
        if (a.prof && tid == 0) khg_atomic_max64(&a.prof->prof_end, khg_clock());
    }
}
