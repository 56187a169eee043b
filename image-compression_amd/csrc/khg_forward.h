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
    KHG_RD_PAD = 4,              // the column pipeline also looks up the row pairs it only flushes (4 before, 4 after)
    KHG_RD_N = KHG_STRIP / 2 + KHG_MAX_SEG_ROWS / 2 + 2 * KHG_RD_PAD  // reciprocal-divisor entries per band
};
static_assert(KHG_MAX_SEG_ROWS <= 254, "a column thread's symbol tallies (KhgHot) are 8-bit fields");

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
    double rdiv[NBAND * KHG_RD_N];  // 1/divisor of band slot s at diagonal u: [s * KHG_RD_N + KHG_RD_PAD + u]
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

// ---- fetching a window.  A window is 4 halo samples, the run's own 16, 4 halo samples.  The own part is loaded
// with 16-byte loads; a halo part that falls off the plane is mirrored (whole-sample reflection reverses the order,
// so it is one unaligned load of the mirror image, reversed in registers).  Parts further out than the 4 samples next
// to the plane feed only outputs that are discarded and are left zero.  Planes narrower than 8 take a per-sample
// path (their reflections can bounce more than once).

KHG_DEV uint32_t khg_ld32u(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }  // any alignment
KHG_DEV uint32_t khg_rot16(uint32_t v) { return (v >> 16) | (v << 16); }

// Luma samples gx..gx+3 (gx = 0 mod 4) of the rows pr/pg/pb into word k of each channel.
KHG_DEV void khg_fwd_fetch4_y(KhgFwdRaw<KHG_SRC_Y>& raw, int k, int gx, int pw, const uint8_t* pr, const uint8_t* pg,
                              const uint8_t* pb)
{
    uint32_t r = 0, g = 0, b = 0;
    if (gx >= 0 && gx + 3 < pw) {  // pw = 0 mod 8: four samples are inside or outside together
        r = khg_ld32(pr + gx); g = khg_ld32(pg + gx); b = khg_ld32(pb + gx);
    } else if (pw >= 8) {
        if (gx == -4 || gx == pw) {  // samples 4,3,2,1 resp. pw-2 .. pw-5
            const int m = gx < 0 ? 1 : pw - 5;
            r = khg_bswap32(khg_ld32u(pr + m)); g = khg_bswap32(khg_ld32u(pg + m)); b = khg_bswap32(khg_ld32u(pb + m));
        }
    } else {
        KHG_OPAQUE(gx);
        for (int i = 0; i < 4; ++i) {
            int p = khg_reflect(gx + i, pw);
            r |= (uint32_t)pr[p] << (8 * i); g |= (uint32_t)pg[p] << (8 * i); b |= (uint32_t)pb[p] << (8 * i);
        }
    }
    raw.w[k] = r; raw.w[6 + k] = g; raw.w[12 + k] = b;
}

// Chroma sites gx..gx+3 (every second byte of the pixel row) into words 2k, 2k+1 of each channel.
template <int SRC>
KHG_DEV void khg_fwd_fetch4_c(KhgFwdRaw<SRC>& raw, int k, int gx, int pw, const uint8_t* pr, const uint8_t* pg,
                              const uint8_t* pb)
{
    uint32_t r[2] = {0, 0}, g[2] = {0, 0}, b[2] = {0, 0};
    if (gx >= 0 && gx + 3 < pw) {  // pw = 0 mod 4
        r[0] = khg_ld32(pr + 2 * gx); r[1] = khg_ld32(pr + 2 * gx + 4);
        g[0] = khg_ld32(pg + 2 * gx); g[1] = khg_ld32(pg + 2 * gx + 4);
        b[0] = khg_ld32(pb + 2 * gx); b[1] = khg_ld32(pb + 2 * gx + 4);
    } else if (pw >= 8) {
        if (gx == -4 || gx == pw) {  // sites m, m-1 | m-2, m-3 with m = 4 resp. pw-2; a word holds two sites 16 bits apart
            const int m = gx < 0 ? 4 : pw - 2;
            r[0] = khg_rot16(khg_ld32u(pr + 2 * m - 2)); r[1] = khg_rot16(khg_ld32u(pr + 2 * m - 6));
            g[0] = khg_rot16(khg_ld32u(pg + 2 * m - 2)); g[1] = khg_rot16(khg_ld32u(pg + 2 * m - 6));
            b[0] = khg_rot16(khg_ld32u(pb + 2 * m - 2)); b[1] = khg_rot16(khg_ld32u(pb + 2 * m - 6));
        }
    } else {
        KHG_OPAQUE(gx);
        for (int i = 0; i < 4; ++i) {
            int p = 2 * khg_reflect(gx + i, pw);
            r[i >> 1] |= (uint32_t)pr[p] << (16 * (i & 1));
            g[i >> 1] |= (uint32_t)pg[p] << (16 * (i & 1));
            b[i >> 1] |= (uint32_t)pb[p] << (16 * (i & 1));
        }
    }
    raw.w[2 * k] = r[0]; raw.w[2 * k + 1] = r[1];
    raw.w[12 + 2 * k] = g[0]; raw.w[12 + 2 * k + 1] = g[1];
    raw.w[24 + 2 * k] = b[0]; raw.w[24 + 2 * k + 1] = b[1];
}

// f64 samples gx, gx+1 (gx even) of a plane row into pair k.
KHG_DEV void khg_fwd_fetch2_p(KhgFwdRaw<KHG_SRC_PLANE>& raw, int k, int gx, int pw, const double* row)
{
    double v[2] = {0, 0};
    if (gx >= 0 && gx + 1 < pw) {  // pw is even: a pair is inside or outside as a whole
        __builtin_memcpy(v, row + gx, 16);  // one 16-byte load
    } else if (pw >= 8) {
        if (gx >= -4 && gx < pw + 4) {  // the mirror image is the pair m-1, m swapped
            const int m = gx < 0 ? -gx : 2 * (pw - 1) - gx;
            double t[2];
            __builtin_memcpy(t, row + m - 1, 16);
            v[0] = t[1]; v[1] = t[0];
        }
    } else {
        KHG_OPAQUE(gx);
        v[0] = row[khg_reflect(gx, pw)];
        v[1] = row[khg_reflect(gx + 1, pw)];
    }
    __builtin_memcpy(&raw.w[4 * k], v, 16);
}

// Row gy of the plane, samples gx0 .. gx0+23 (gx0 = -4 mod 16; columns outside [0, pw) are reflected).
template <int SRC>
KHG_DEV void khg_fwd_fetch(const KhgFwdArgs& a, KhgFwdRaw<SRC>& raw, int gy, int gx0)
{
    const int c0 = gx0 + 4;                  // the run's own samples: c0 .. c0+15
    const bool own = c0 + 16 <= a.pw;        // all inside
    if constexpr (SRC == KHG_SRC_PLANE) {
        const double* row = a.src + (size_t)gy * a.src_pitch;
        KHG_UNROLL
        for (int k = 0; k < 12; ++k) {
            if (k >= 2 && k < 10 && own)
                __builtin_memcpy(&raw.w[4 * k], row + c0 + 2 * (k - 2), 16);
            else
                khg_fwd_fetch2_p(raw, k, gx0 + 2 * k, a.pw, row);
        }
    } else if constexpr (SRC == KHG_SRC_Y) {
        const size_t rowoff = (size_t)khg_band_row(a, gy) * a.img_w;
        const uint8_t *pr = a.r + rowoff, *pg = a.g + rowoff, *pb = a.b + rowoff;
        if (own) {
            __builtin_memcpy(&raw.w[1], pr + c0, 16);
            __builtin_memcpy(&raw.w[7], pg + c0, 16);
            __builtin_memcpy(&raw.w[13], pb + c0, 16);
        } else {
            for (int k = 1; k < 5; ++k) khg_fwd_fetch4_y(raw, k, gx0 + 4 * k, a.pw, pr, pg, pb);
        }
        khg_fwd_fetch4_y(raw, 0, gx0, a.pw, pr, pg, pb);
        khg_fwd_fetch4_y(raw, 5, gx0 + 20, a.pw, pr, pg, pb);
    } else {
        // chroma site (gy, gx) reads the pixel at (2gy, 2gx): Colorspace.h:16-21
        const size_t rowoff = (size_t)khg_band_row(a, 2 * gy) * a.img_w;
        const uint8_t *pr = a.r + rowoff, *pg = a.g + rowoff, *pb = a.b + rowoff;
        if (own) {
            __builtin_memcpy(&raw.w[2], pr + 2 * c0, 32);
            __builtin_memcpy(&raw.w[14], pg + 2 * c0, 32);
            __builtin_memcpy(&raw.w[26], pb + 2 * c0, 32);
        } else {
            for (int k = 1; k < 5; ++k) khg_fwd_fetch4_c<SRC>(raw, k, gx0 + 4 * k, a.pw, pr, pg, pb);
        }
        khg_fwd_fetch4_c<SRC>(raw, 0, gx0, a.pw, pr, pg, pb);
        khg_fwd_fetch4_c<SRC>(raw, 5, gx0 + 20, a.pw, pr, pg, pb);
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
// in a register keeps 64 lanes from hammering one LDS word.  Bits 0-23: three 8-bit counters (a column thread
// emits at most KHG_MAX_SEG_ROWS symbols per segment); bits 24-29: one `seen` bit per (symbol, output stream).
// A thread's raster indices rise within each of its two output streams (rows k and rows hh+k), so only the
// first hit of a symbol in a stream can lower the workgroup's first-appearance index: one LDS atomic per bit, ever.
typedef uint32_t KhgHot;

// Quantise one coefficient, store the symbol and count it (Compress.h:218-219, Huffman.h:110-139).
// rdiv = RN(1/divisor) of this position (khg_round_quot_fast); stream 0 = rows k, 1 = rows hh+k.
template <class SMEM>
KHG_DEV void khg_quant_store(const KhgFwdArgs& a, SMEM& sh, KhgHot& hot, double c, double rdiv, int band,
                             int stream, int y, int x)
{
    double n;
    if (!khg_round_quot_fast(c, rdiv, &n)) {
        // near a tie, not finite, or huge: the reference's own division decides
        n = round(c / khg_divisor(a.q, band, x, y, a.lvl));
        if (!(n >= -(double)KHG_SYM_BIAS && n < (double)KHG_SYM_BIAS)) n = (double)KHG_SYM_BIAS;  // flagged below
    }
    int q = (int)n;  // saturating: a finite n beyond int range stays out of range
    if ((unsigned)(q + KHG_SYM_BIAS) >= 2u * KHG_SYM_BIAS) {
        khg_set_status(a.status, -4);
        q = 0;
    }
    uint32_t raster = (uint32_t)y * (uint32_t)a.sym_pitch + (uint32_t)x;
    a.sym[raster] = (int16_t)q;
    const uint32_t q1 = (uint32_t)(q + 1);
    if (q1 <= 2u) {
        hot += 1u << (8 * q1);
        const uint32_t bit = 1u << (24 + 2 * q1 + stream);
        if (!(hot & bit)) {
            hot |= bit;
            khg_lds_min(&sh.first[q + SMEM::LHIST_BIAS], raster);
        }
    } else {
        const int l = q + SMEM::LHIST_BIAS;
        if ((unsigned)l < (unsigned)SMEM::LHIST) {
            khg_lds_add(&sh.cnt[l], 1u);
            khg_lds_min(&sh.first[l], raster);
        } else {
            khg_atomic_add(&a.hist_cnt[q + KHG_SYM_BIAS], 1u);
            khg_atomic_min(&a.hist_first[q + KHG_SYM_BIAS], raster);
        }
    }
}

// What a column-phase thread knows about its column for the whole segment.
struct KhgColCtx {
    int half, tt;  // de-interleaved half (0: even source columns) and index within it
    int xq, xo;    // column within a quadrant, packed output column
    int hh, k0, k1;
    bool colok;
};

// Finished, unscaled pair (s2, d2) of row pair k of one column: scale, store / quantise (rl, rh: reciprocal
// divisors of the low-row and high-row band at this position).
template <int LEVEL, class SMEM>
KHG_DEV void khg_col_emit(const KhgFwdArgs& a, SMEM& sh, KhgHot& hot, const KhgColCtx& cx, int k, double s2,
                          double d2, double rl, double rh)
{
    if (!(cx.colok && k >= cx.k0 && k < cx.k1)) return;
    const double lo = khg_fwd_scale_even(s2), hi = khg_fwd_scale_odd(d2);
    if (a.coef) {
        a.coef[(size_t)k * a.coef_pitch + cx.xo] = lo;
        a.coef[(size_t)(cx.hh + k) * a.coef_pitch + cx.xo] = hi;
    }
    const bool emit = k >= a.emit_k0 && k < a.emit_k1;
    if (LEVEL == 1) {
        if (cx.half == 0) {
            a.ll[(size_t)k * a.ll_pitch + cx.xq] = lo;
            if (emit) khg_quant_store(a, sh, hot, hi, rh, 5, 1, cx.hh + k, cx.xo);
        } else if (emit) {
            khg_quant_store(a, sh, hot, lo, rl, 4, 0, k, cx.xo);
            khg_quant_store(a, sh, hot, hi, rh, 6, 1, cx.hh + k, cx.xo);
        }
    } else if (emit) {
        khg_quant_store(a, sh, hot, lo, rl, cx.half, 0, k, cx.xo);
        khg_quant_store(a, sh, hot, hi, rh, 2 + cx.half, 1, cx.hh + k, cx.xo);
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
    const int flip = __builtin_popcount((unsigned)bid >> 3) & 1;  // see the column phase

    KhgLift cs[KHG_NSLOT(256)];
    KhgHot hots[KHG_NSLOT(256)];
    KhgFwdRaw<SRC> raws[KHG_NSLOT(256)];
    KHG_FOR_THREADS(256)
    {
        if (a.prof && tid == 0) khg_atomic_max64(&a.prof->prof_not_start, ~khg_clock());
        KhgLift z = {0, 0, 0, 0, 0};
        cs[KHG_SLOT] = z;
        hots[KHG_SLOT] = 0;
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
            sh.rdiv[s * KHG_RD_N + KHG_RD_PAD + u] = 1.0 / khg_divisor(a.q, band, bx + (x0 >> 1) + u, by + k0, a.lvl);
        }
    }
    KHG_BARRIER();

    // developer build (-DKHG_PHASE_TIMING): core-clock time three workgroups spend per phase, printed at the end
#if defined(KHG_PHASE_TIMING) && !defined(KHG_SIM)
    long long tacc[5] = {0, 0, 0, 0, 0}, tprev = clock64();
#define KHG_PT(i) { long long t_ = clock64(); tacc[i] += t_ - tprev; tprev = t_; }
#else
#define KHG_PT(i)
#endif
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
            KHG_PT(0)
            // next batch's window: in flight while the column phase runs
            if (b + 1 < nb)
                khg_fwd_fetch<SRC>(a, raw, khg_reflect(rin0 + (b + 1) * KHG_BATCH + rr, a.ph), x0 + run * 16 - 4);
            KHG_PT(1)
        }
        KHG_BARRIER();
        KHG_PT(2)
        KHG_FOR_THREADS(256)
        {
            // ---- column phase: one de-interleaved column per thread.  Which half the first two waves take
            // alternates between workgroups: the odd half quantises two bands, the even half one, and a
            // workgroup's wave w always lands on the same SIMD of its CU.
            const int col = tid ^ (flip << 7);
            KhgColCtx cx;
            cx.half = col >> 7;
            cx.tt = col & 127;
            cx.colok = x0 + 2 * cx.tt + cx.half < a.pw;
            cx.xq = (x0 >> 1) + cx.tt;
            cx.xo = cx.half ? hw + cx.xq : cx.xq;
            cx.hh = hh; cx.k0 = k0; cx.k1 = k1;
            // slots of the two bands this thread quantises (level 1, even half: the high-row band only)
            const int slot_l = LEVEL == 1 ? 0 : cx.half, slot_h = LEVEL == 1 ? 1 + cx.half : 2 + cx.half;
            const int kb = (rin0 + b * KHG_BATCH) / 2 - 2;  // row pair the first step of this batch finishes
            const double* rdl = &sh.rdiv[slot_l * KHG_RD_N + KHG_RD_PAD + cx.tt + (kb - k0)];
            const double* rdh = &sh.rdiv[slot_h * KHG_RD_N + KHG_RD_PAD + cx.tt + (kb - k0)];
            const double* bp = &sh.B[col];
            KhgLift st = cs[KHG_SLOT];
            KhgHot hot = hots[KHG_SLOT];
            // Two row pairs per trip, every LDS operand of the trip requested up front: one wait per trip instead
            // of one per use.  Kept as a loop: unrolled further, the rarely taken division paths of the quantiser
            // alone make the kernel larger than the instruction cache likes.
            KHG_NOUNROLL
            for (int j = 0; j < KHG_BATCH / 2; j += 2) {
                const double e0 = bp[(2 * j) * KHG_B_PITCH], o0 = bp[(2 * j + 1) * KHG_B_PITCH];
                const double e1 = bp[(2 * j + 2) * KHG_B_PITCH], o1 = bp[(2 * j + 3) * KHG_B_PITCH];
                const double rl0 = rdl[j], rl1 = rdl[j + 1], rh0 = rdh[j], rh1 = rdh[j + 1];
                double s2, d2;
                khg_fwd_step(st, e0, o0, s2, d2);
                khg_col_emit<LEVEL>(a, sh, hot, cx, kb + j, s2, d2, rl0, rh0);
                khg_fwd_step(st, e1, o1, s2, d2);
                khg_col_emit<LEVEL>(a, sh, hot, cx, kb + j + 1, s2, d2, rl1, rh1);
            }
            cs[KHG_SLOT] = st;
            hots[KHG_SLOT] = hot;
            KHG_PT(3)
        }
        KHG_BARRIER();
        KHG_PT(4)
    }
#if defined(KHG_PHASE_TIMING) && !defined(KHG_SIM)
    if (SRC == KHG_SRC_Y && (bid == 0 || bid == 333 || bid == 700) && (threadIdx.x & 63) == 0)
        printf("PT bid %d wave %d nb %d: row %lld fetch %lld bar %lld col %lld bar %lld\n", bid, (int)threadIdx.x >> 6, nb,
               tacc[0] / nb, tacc[1] / nb, tacc[2] / nb, tacc[3] / nb, tacc[4] / nb);
#endif

    KHG_FOR_THREADS(256)
    {
        const KhgHot hot = hots[KHG_SLOT];
        for (int v = 0; v < 3; ++v) {
            uint32_t c = (hot >> (8 * v)) & 0xFFu;
            if (c) khg_lds_add(&sh.cnt[KhgFwdSmem<LEVEL>::LHIST_BIAS - 1 + v], c);
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
