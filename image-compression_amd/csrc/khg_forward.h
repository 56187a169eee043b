// khg_forward.h -- colour transform + one 2-D CDF 9/7 lifting level + quantiser + histogram.
//
// Lane-streaming layout.  A wave owns a strip of 64 sample pairs (128 columns) of the plane being transformed and
// walks down a segment of its rows, two rows per trip; lane l holds the pair (x[2p], x[2p+1]), p = first pair + l:
//   rows   : the four lifting steps of a row need the neighbouring pair's value once each (e[p+1], d1[p-1],
//            s1[p+1], d2[p-1]); those come from the neighbouring LANE by a DPP wave shift, so a row is transformed
//            where it was loaded, with no window recomputed per thread, no staging tile and no barrier.  The two
//            pairs at each end of a strip are halo (their results are wrong and not used): 60 of 64 pairs are output.
//            Halo pairs that fall outside the plane are LOADED mirrored (whole-sample symmetric extension about the
//            first and the last sample: x[-i] = x[i], x[n-1+i] = x[n-1-i]).  Lifting preserves that symmetry and
//            a + b == b + a bit for bit, so the interior formula then yields exactly what the reference's edge
//            lines compute (DWT.h:18,26,34,42: x[n-1] += 2*a*x[n-2], x[0] += 2*a*x[1], and 2*a*x == a*(x+x)), and
//            the row pass has no edge case at all.  A mirrored pair is the memory pair (x[i], x[i+1]) read backwards,
//            so a lane carries its load column and which of the two samples is its even one.
//   columns: the lane's two packed output columns (low half p, high half w/2 + p) each run the streaming lifting
//            pipeline of khg_fwd_step over the rows as they arrive, state in registers.  Finished coefficients are
//            scaled and emitted at once: level-1 LL to an f64 buffer (input of level 2), everything else quantised
//            to int16 symbols in Mallat layout and counted (count + first raster index).
// Sources: u8 RGB -> Y (Main.cpp:200-202, Colorspace.h:11), u8 RGB -> Cb and Cr together (one load of the pixel feeds
// both planes; Colorspace.h:15-22 decimation: the pixel at even row, even column), or an f64 plane.
// Arithmetic order follows DWT.h:7-60 exactly (no FMA contraction: build with -ffp-contract=off).
#pragma once
#include "khg_common.h"

// Sources of a transform: planar u8 R, G, B -> Y / -> Cb and Cr; an f64 plane; and the pixel array of a 24-bit BMP
// (B, G, R bytes per pixel, rows bottom-up, KhuGleSignal.cpp:264-272) -> Y / -> Cb and Cr.
enum { KHG_SRC_Y = 0, KHG_SRC_C = 1, KHG_SRC_PLANE = 3, KHG_SRC_Y_BGR = 4, KHG_SRC_C_BGR = 5 };
#define KHG_SRC_IS_LUMA(SRC) ((SRC) == KHG_SRC_Y || (SRC) == KHG_SRC_Y_BGR)
#define KHG_SRC_IS_BGR(SRC) ((SRC) == KHG_SRC_Y_BGR || (SRC) == KHG_SRC_C_BGR)

enum {
    KHG_LANES = 64,
    KHG_STRIP_PAIRS = 60,        // output pairs per wave strip (64 lanes minus two halo pairs per side)
    KHG_WG_STRIPS = 4,           // waves (adjacent strips) per workgroup
    KHG_MAX_SEG_ROWS = 120,      // rows per workgroup segment, at most
    KHG_RD_PAD = 4,              // the column pipeline also looks up the row pairs it only flushes
    KHG_RD_N = KHG_WG_STRIPS * KHG_STRIP_PAIRS + 4 + KHG_MAX_SEG_ROWS / 2 + 2 * KHG_RD_PAD  // reciprocals per band
};
static_assert(4 * (KHG_MAX_SEG_ROWS / 2) <= 255, "a lane's symbol tallies (KhgHot) are 8-bit fields");

// One plane of a launch.
struct KhgFwdPlane {
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
    int row_begin, row_end;    // plane rows this launch transforms (even; the whole plane, or one tile's band)
    int emit_k0, emit_k1;      // row pairs whose detail symbols are stored and counted (a tile recomputes its
                               // neighbours' edge rows for the LL quadrant only)
    int seg_rows;              // rows of the tallest workgroup segment (even; see khg_segment_rows)
    int nseg;                  // segments
    int nstrip_wg;             // workgroups side by side (KHG_WG_STRIPS strips each)
    int wg0;                   // first workgroup of this plane in the launch
    KhgQuant q;
};

struct KhgFwdArgs {
    const uint8_t *r, *g, *b;  // full-resolution RGB planes, or a band of their rows; BGR sources: r = the pixel array
    int img_w;                 // image columns (the planes' row pitch)
    int bgr_stride;            // BGR sources: bytes per pixel row (3 * img_w rounded up to a multiple of 4)
    int in_row0;               // image row held by the first row of r/g/b (0 for a whole image)
    int in_rows;               // image rows r/g/b hold
    int* status;
    KhgResult* prof;           // when set, the launch's first start / last end device clock go here
    int lvl;                   // preset 0..2
    int nplanes;
    int nwg;                   // workgroups of the launch
    // RGB launch: [0] = Y, [1] = Cb, [2] = Cr (Cb and Cr share one grid: a lane carries both).
    // Plane launches: up to three independent planes.
    KhgFwdPlane pl[3];
    int lhist;                 // level 2: bins of the workgroup histogram in use (a power of two up to the capacity)
};

// LDS of one workgroup.  Level 1 emits detail bands only (small symbols, three bands): 256 bins.  Level 2 also emits
// the LL band, whose symbols spread over hundreds of values and at the Normal preset (LL step 1.011) reach 1009: room
// for 2048 bins (more would cost the fourth workgroup per CU that the launch plan counts on), of which a launch uses
// KhgFwdArgs::lhist.  Symbols outside the bins go to the global histogram one atomic at a time, and neighbouring LL
// symbols collide there: with 1024 bins the Normal preset's level 2 took 0.75 ms instead of 0.05.
template <int LEVEL, int NP>
struct KhgFwdSmem {
    enum { LHIST = LEVEL == 1 ? 256 : 2048, LHIST_BIAS = LHIST / 2, NBAND = LEVEL == 1 ? 3 : 4 };
    uint32_t cnt[NP][LHIST];    // symbols [-LHIST/2, LHIST/2)
    uint32_t first[NP][LHIST];
    double rdiv[NBAND * KHG_RD_N];  // 1/divisor of band slot s at diagonal u: [s * KHG_RD_N + KHG_RD_PAD + u]
};
union KhgFwdSmemAny {
    KhgFwdSmem<1, 1> y;
    KhgFwdSmem<1, 2> c;
    KhgFwdSmem<2, 1> l2;
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
#define KHG_P1 (-1.586134342)
#define KHG_U1 (-0.05298011854)
#define KHG_P2 (0.8829110762)
#define KHG_U2 (0.4435068522)
KHG_DEV void khg_fwd_step(KhgLift& st, double en, double on, double& s2, double& d2)
{
    double d1n = st.o + KHG_P1 * (st.e + en);
    double s1n = st.e + KHG_U1 * (st.d1 + d1n);
    double d2n = st.d1 + KHG_P2 * (st.s1 + s1n);
    double s2n = st.s1 + KHG_U2 * (st.d2 + d2n);
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

// ---- neighbour lanes.  khg_lane_next(v): the value lane l+1 holds, khg_lane_prev(v): lane l-1's.  What the lane
// at the end of a wave receives is unspecified (whatever its destination register held): that lane is halo.
#ifndef KHG_SIM
__device__ __forceinline__ double khg_lane_next(double v)
{
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x130, 0xf, 0xf, false);  // wave_shl:1
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x130, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double khg_lane_prev(double v)
{
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x138, 0xf, 0xf, false);  // wave_shr:1
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
// One exchange of a per-thread value: dst = src of the next / previous lane.
#define KHG_LANE_NEXT(dst, src) KHG_FOR_THREADS(256) { (dst)[KHG_SLOT] = khg_lane_next((src)[KHG_SLOT]); }
#define KHG_LANE_PREV(dst, src) KHG_FOR_THREADS(256) { (dst)[KHG_SLOT] = khg_lane_prev((src)[KHG_SLOT]); }
// A value that is the same in every lane of the wave, told to the compiler (it then lives in a scalar register
// and conditions on it become scalar branches).
#define KHG_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)
KHG_DEV int khg_f2i(double n) { return (int)n; }  // v_cvt_i32_f64 saturates
#else
#define KHG_LANE_NEXT(dst, src) \
    for (int t_ = 0; t_ < 256; ++t_) (dst)[t_] = (t_ & 63) == 63 ? (src)[t_] : (src)[t_ + 1];
#define KHG_LANE_PREV(dst, src) \
    for (int t_ = 0; t_ < 256; ++t_) (dst)[t_] = (t_ & 63) == 0 ? (src)[t_] : (src)[t_ - 1];
#define KHG_UNIFORM(x) (x)
static inline int khg_f2i(double n) { return n >= 2147483647.0 ? 2147483647 : (n <= -2147483648.0 ? (-2147483647 - 1) : (int)n); }
#endif

// Image row `row` as a row of the r/g/b buffers.  Rows a segment reads beyond those its outputs need may, for a band
// of rows (a tile), lie outside the band; they are clamped into it -- their results are discarded either way.
KHG_DEV int khg_band_row(const KhgFwdArgs& a, int row)
{
    int r = row - a.in_row0;
    return r < 0 ? 0 : (r >= a.in_rows ? a.in_rows - 1 : r);
}

// ---- source of one lane for one row pair, as loaded (converted when the rows are transformed, one trip later,
// so the loads fly during the previous trip's arithmetic).
template <int SRC>
struct KhgFwdRaw {
    // Y: r, g, b of row a, then of row b, two pixels per 16-bit load.  C: r, g, b of the two pixel rows, four
    // pixels per 32-bit load (bytes 0 and 2 are the two chroma sites).  PLANE: the pair of row a, the pair of row b.
    // Y_BGR: the six bytes of two pixels as three 16-bit loads per row.  C_BGR: two 32-bit loads per row, one per site.
    uint32_t w[SRC == KHG_SRC_PLANE ? 8 : 6];
};

KHG_DEV uint32_t khg_ld16(const uint8_t* p) { uint16_t v; __builtin_memcpy(&v, p, 2); return v; }
KHG_DEV uint32_t khg_ld32(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }

// Offset (in elements of the source) of plane row y's first sample: a wave-uniform number.
template <int SRC>
KHG_DEV uint32_t khg_fwd_row_offset(const KhgFwdArgs& a, const KhgFwdPlane& pl, int y)
{
    if (SRC == KHG_SRC_PLANE) return (uint32_t)y * (uint32_t)pl.src_pitch;
    // chroma site (y, x) reads the pixel at (2y, 2x): Colorspace.h:16-21
    const int r = khg_band_row(a, KHG_SRC_IS_LUMA(SRC) ? y : 2 * y);
    if (KHG_SRC_IS_BGR(SRC)) return (uint32_t)(a.in_rows - 1 - r) * (uint32_t)a.bgr_stride;  // the file's rows run bottom-up
    return (uint32_t)r * (uint32_t)a.img_w;
}

// Offsets of the two source rows of row pair j (plane rows 2j and 2j+1, reflected into the plane).  Nearly every
// trip's rows are plain rows of the plane and of the r/g/b band: that case is a few scalar instructions.
template <int SRC>
KHG_DEV void khg_fwd_row_offsets(const KhgFwdArgs& a, const KhgFwdPlane& pl, int j, uint32_t& oa, uint32_t& ob)
{
    const int ya = 2 * j, ph = KHG_UNIFORM(pl.ph);
    if (ya >= 0 && ya + 1 < ph) {
        if (SRC == KHG_SRC_PLANE) {
            oa = (uint32_t)ya * (uint32_t)KHG_UNIFORM(pl.src_pitch);
            ob = oa + (uint32_t)KHG_UNIFORM(pl.src_pitch);
            return;
        }
        const int f = KHG_SRC_IS_LUMA(SRC) ? 1 : 2;
        const int ra = f * ya - KHG_UNIFORM(a.in_row0), rb = ra + f;
        if (ra >= 0 && rb < KHG_UNIFORM(a.in_rows)) {
            if (KHG_SRC_IS_BGR(SRC)) {
                oa = (uint32_t)(KHG_UNIFORM(a.in_rows) - 1 - ra) * (uint32_t)KHG_UNIFORM(a.bgr_stride);
                ob = (uint32_t)(KHG_UNIFORM(a.in_rows) - 1 - rb) * (uint32_t)KHG_UNIFORM(a.bgr_stride);
            } else {
                oa = (uint32_t)ra * (uint32_t)KHG_UNIFORM(a.img_w);
                ob = (uint32_t)rb * (uint32_t)KHG_UNIFORM(a.img_w);
            }
            return;
        }
    }
    oa = KHG_UNIFORM(khg_fwd_row_offset<SRC>(a, pl, khg_reflect(ya, ph)));
    ob = KHG_UNIFORM(khg_fwd_row_offset<SRC>(a, pl, khg_reflect(ya + 1, ph)));
}

// Where a lane's pair (x[2p], x[2p+1]) is read from: both samples reflected into the row (p may lie up to two pairs
// outside the plane, and anywhere for lanes beyond the last strip, whose loads are not used).  The two reflected
// columns are always neighbours; `col` is the lower one and `swap` says that it holds the odd sample.
struct KhgLaneSrc {
    uint32_t col;  // column offset of the load in elements of the source (planar RGB: pixels; BGR: bytes)
    uint32_t she;  // planar RGB: bit position of the even sample's byte in the loaded word; others: nonzero = swapped
    uint32_t sho;  // planar RGB: bit position of the odd sample's byte
};

template <int SRC>
KHG_DEV KhgLaneSrc khg_fwd_lane_src(int p, int pw)
{
    const int ec = khg_reflect(2 * p, pw), oc = khg_reflect(2 * p + 1, pw);
    const bool swap = ec > oc;
    const uint32_t lo = (uint32_t)(swap ? oc : ec);
    const uint32_t sh = SRC == KHG_SRC_Y ? 8u : 16u;  // distance of the pair's samples in the word: chroma sites are
    KhgLaneSrc l;                                     // every second pixel (Colorspace.h:16-21)
    l.col = SRC == KHG_SRC_C ? 2u * lo : SRC == KHG_SRC_Y_BGR ? 3u * lo : SRC == KHG_SRC_C_BGR ? 6u * lo : lo;
    l.she = swap ? sh : 0u;
    l.sho = swap ? 0u : sh;
    return l;
}

// Two rows of the plane at row offsets oa, ob (khg_fwd_row_offsets), at the lane's column.  Offsets are 32-bit: a
// plane has fewer than 2^31 samples.  Loads of 2 / 4 / 16 bytes at any 1 / 2 / 8-byte boundary.
template <int SRC>
KHG_DEV void khg_fwd_fetch(const KhgFwdArgs& a, const KhgFwdPlane& pl, KhgFwdRaw<SRC>& raw, uint32_t oa, uint32_t ob,
                           uint32_t col)
{
    oa += col;
    ob += col;
    if constexpr (SRC == KHG_SRC_PLANE) {
        __builtin_memcpy(&raw.w[0], pl.src + oa, 16);
        __builtin_memcpy(&raw.w[4], pl.src + ob, 16);
    } else if constexpr (SRC == KHG_SRC_Y) {
        raw.w[0] = khg_ld16(a.r + oa); raw.w[1] = khg_ld16(a.g + oa); raw.w[2] = khg_ld16(a.b + oa);
        raw.w[3] = khg_ld16(a.r + ob); raw.w[4] = khg_ld16(a.g + ob); raw.w[5] = khg_ld16(a.b + ob);
    } else if constexpr (SRC == KHG_SRC_Y_BGR) {  // B G R B G R of two neighbouring pixels
        raw.w[0] = khg_ld16(a.r + oa); raw.w[1] = khg_ld16(a.r + oa + 2); raw.w[2] = khg_ld16(a.r + oa + 4);
        raw.w[3] = khg_ld16(a.r + ob); raw.w[4] = khg_ld16(a.r + ob + 2); raw.w[5] = khg_ld16(a.r + ob + 4);
    } else if constexpr (SRC == KHG_SRC_C_BGR) {  // B G R (and one more byte) of two pixels two columns apart
        raw.w[0] = khg_ld32(a.r + oa); raw.w[1] = khg_ld32(a.r + oa + 6);
        raw.w[2] = khg_ld32(a.r + ob); raw.w[3] = khg_ld32(a.r + ob + 6);
    } else {
        raw.w[0] = khg_ld32(a.r + oa); raw.w[1] = khg_ld32(a.g + oa); raw.w[2] = khg_ld32(a.b + oa);
        raw.w[3] = khg_ld32(a.r + ob); raw.w[4] = khg_ld32(a.g + ob); raw.w[5] = khg_ld32(a.b + ob);
    }
}

// Samples (even, odd) of row rr (0: a, 1: b) of channel n (Y / plane: 0; chroma: 0 = Cb, 1 = Cr) as the f64 values
// the transform runs on.  Main.cpp:200-202 widens u8 to double; Colorspace.h:11,19,20: products and sums left to right.
template <int SRC>
KHG_DEV void khg_fwd_samples(const KhgFwdRaw<SRC>& raw, const KhgLaneSrc& ls, int rr, int n, double& e, double& o)
{
    if constexpr (SRC == KHG_SRC_PLANE) {
        double lo, hi;
        __builtin_memcpy(&lo, &raw.w[4 * rr], 8);
        __builtin_memcpy(&hi, &raw.w[4 * rr + 2], 8);
        e = ls.she ? hi : lo;
        o = ls.she ? lo : hi;
    } else if constexpr (KHG_SRC_IS_BGR(SRC)) {
        uint32_t pa, pb;  // (B, G, R) of the lower and of the upper pixel in bits 0-23
        if (SRC == KHG_SRC_Y_BGR) {
            const uint32_t w0 = raw.w[3 * rr], w1 = raw.w[3 * rr + 1], w2 = raw.w[3 * rr + 2];
            pa = w0 | (w1 << 16);
            pb = (w1 >> 8) | (w2 << 8);
        } else {
            pa = raw.w[2 * rr];
            pb = raw.w[2 * rr + 1];
        }
        const double B0 = (double)(pa & 255u), G0 = (double)((pa >> 8) & 255u), R0 = (double)((pa >> 16) & 255u);
        const double B1 = (double)(pb & 255u), G1 = (double)((pb >> 8) & 255u), R1 = (double)((pb >> 16) & 255u);
        double va, vb;
        if (SRC == KHG_SRC_Y_BGR) {
            va = R0 * 0.299 + G0 * 0.587 + B0 * 0.114;
            vb = R1 * 0.299 + G1 * 0.587 + B1 * 0.114;
        } else if (n == 0) {
            va = R0 * -0.16874 + G0 * -0.33126 + B0 * 0.5;
            vb = R1 * -0.16874 + G1 * -0.33126 + B1 * 0.5;
        } else {
            va = R0 * 0.5 + G0 * -0.41869 + B0 * -0.08131;
            vb = R1 * 0.5 + G1 * -0.41869 + B1 * -0.08131;
        }
        e = ls.she ? vb : va;
        o = ls.she ? va : vb;
    } else {
        const uint32_t r = raw.w[3 * rr], g = raw.w[3 * rr + 1], b = raw.w[3 * rr + 2];
        const double R0 = (double)((r >> ls.she) & 255u), G0 = (double)((g >> ls.she) & 255u),
                     B0 = (double)((b >> ls.she) & 255u);
        const double R1 = (double)((r >> ls.sho) & 255u), G1 = (double)((g >> ls.sho) & 255u),
                     B1 = (double)((b >> ls.sho) & 255u);
        if (SRC == KHG_SRC_Y) {
            e = R0 * 0.299 + G0 * 0.587 + B0 * 0.114;
            o = R1 * 0.299 + G1 * 0.587 + B1 * 0.114;
        } else if (n == 0) {
            e = R0 * -0.16874 + G0 * -0.33126 + B0 * 0.5;
            o = R1 * -0.16874 + G1 * -0.33126 + B1 * 0.5;
        } else {
            e = R0 * 0.5 + G0 * -0.41869 + B0 * -0.08131;
            o = R1 * 0.5 + G1 * -0.41869 + B1 * -0.08131;
        }
    }
}

// Per-thread tally of the symbols -1, 0, +1, which make up almost all detail-band symbols: counting them
// in a register keeps 64 lanes from hammering one LDS word.  Bits 0-23: three 8-bit counters (a lane emits at most
// four symbols per row pair of a segment), symbol q in byte 1 - q; bits 24-29: one `seen` bit per (symbol, output
// stream), symbol q of stream t at bit 24 + 3 t + q + 1.  Both layouts make a symbol's word one instruction.
// A lane's raster indices rise within each of its two output streams (rows k: low-half column, then high-half
// column; rows hh+k: the same), so only the first hit of a symbol in a stream can lower the workgroup's
// first-appearance index: one LDS atomic per bit, ever.
typedef uint32_t KhgHot;

// ---- quantiser (Compress.h:218-219) and symbol table counts (Huffman.h:110-139).
// The symbols a lane emits for one row pair are quantised together: the common case -- every quotient is clear of a
// rounding tie and every symbol is -1, 0 or +1 and has been seen in its stream before -- is straight-line code, and
// everything else is collected behind two rarely taken branches.

// q = round(c / div) from rdiv = RN(1/div), when that is safe: t = c * rdiv is within 2^-51 (relative) of the exact
// quotient, far less than 1e-5 for every |t| that can become a symbol, so both round to the same integer unless
// t sits within 1e-5 of a half-integer (then, and for NaN and infinity, the caller divides).  Away from ties
// round-to-nearest-even and round-half-away agree.  A huge finite t passes and saturates q, which is out of the
// symbol range and fails the job like the division would.
KHG_DEV bool khg_quant_try(double c, double rdiv, int* q)
{
    const double t = c * rdiv;
    const double n = __builtin_rint(t);
    *q = khg_f2i(n);
    return fabs(t - n) < 0.49999;
}

// The reference's own division, for the coefficients khg_quant_try declines.
KHG_DEV int khg_quant_divide(const KhgFwdArgs& a, const KhgFwdPlane& pl, double c, int band, int x, int y)
{
    double n = round(c / khg_divisor(pl.q, band, x, y, a.lvl));
    if (!(n >= -(double)KHG_SYM_BIAS && n < (double)KHG_SYM_BIAS)) n = (double)KHG_SYM_BIAS;  // out of range: flagged later
    return (int)n;
}

// Tally words of a hot symbol q = q1 - 1 in -1..1: its 8-bit counter's unit, and its seen bit in output stream
// `stream`.  For other q1 the results are meaningless (and unused).
#ifndef KHG_SIM
KHG_DEV uint32_t khg_hot_unit(uint32_t q1) { return __builtin_amdgcn_alignbyte(0u, 0x10000u, q1); }  // 0x10000 >> 8 (q1 & 3)
#else
KHG_DEV uint32_t khg_hot_unit(uint32_t q1) { return 0x10000u >> (8u * (q1 & 3u)); }
#endif
KHG_DEV uint32_t khg_hot_seen(uint32_t q1, int stream) { return (1u << (24 + 3 * stream)) << (q1 & 31u); }

// Count one symbol of a row pair that the straight-line path could not finish (it holds a symbol other than -1, 0,
// +1, or the first appearance of one of those in its stream).  Returns the symbol to store (0 when out of range).
template <int LHIST>
KHG_DEV int khg_count_rare(const KhgFwdArgs& a, const KhgFwdPlane& pl, uint32_t* lcnt, uint32_t* lfirst, KhgHot& hot, int q,
                           int stream, uint32_t raster)
{
    const int lh = LHIST == 256 ? 256 : a.lhist;  // bins in use, centred in the LHIST there is room for
    const uint32_t q1 = (uint32_t)(q + 1);
    if (q1 <= 2u) {
        hot += khg_hot_unit(q1);
        const uint32_t bit = khg_hot_seen(q1, stream);
        if (!(hot & bit)) {
            hot |= bit;
            khg_lds_min(&lfirst[q + LHIST / 2], raster);
        }
        return q;
    }
    if ((unsigned)(q + KHG_SYM_BIAS) >= 2u * KHG_SYM_BIAS) {
        khg_set_status(a.status, -4);
        q = 0;  // stored and counted as a zero; the job fails with KHG_ERR_RANGE
        hot += khg_hot_unit(1u);
        const uint32_t bit = khg_hot_seen(1u, stream);
        if (!(hot & bit)) {
            hot |= bit;
            khg_lds_min(&lfirst[LHIST / 2], raster);
        }
        return q;
    }
    const int l = q + LHIST / 2;
    if ((unsigned)(q + lh / 2) < (unsigned)lh) {
        khg_lds_add(&lcnt[l], 1u);
        khg_lds_min(&lfirst[l], raster);
    } else {
        khg_atomic_add(&pl.hist_cnt[q + KHG_SYM_BIAS], 1u);
        khg_atomic_min(&pl.hist_first[q + KHG_SYM_BIAS], raster);
    }
    return q;
}

// NS coefficients c[i] of one lane and row pair, in raster order within each output stream: band[i], reciprocal
// divisor rd[i], position (y[i], x[i]) -- used only off the fast path -- and byte offset so + boff[i] in the symbol
// plane.  Quantises, stores and counts them.
template <int NS, int LHIST>
KHG_DEV void khg_quant_emit(const KhgFwdArgs& a, const KhgFwdPlane& pl, uint32_t* lcnt, uint32_t* lfirst, KhgHot& hot,
                            const double* c, const double* rd, const int* band, const int* stream, const int* y,
                            const int* x, uint32_t so, const uint32_t* boff)
{
    int q[NS];
    bool clear = true;
    KHG_UNROLL
    for (int i = 0; i < NS; ++i) clear = khg_quant_try(c[i], rd[i], &q[i]) && clear;
    if (!clear) {  // near a tie, not finite: the reference's own division decides
        KHG_UNROLL
        for (int i = 0; i < NS; ++i) {
            int qi;
            if (!khg_quant_try(c[i], rd[i], &qi)) q[i] = khg_quant_divide(a, pl, c[i], band[i], x[i], y[i]);
        }
    }
    uint32_t units = 0, seen = 0, top = 0;
    KHG_UNROLL
    for (int i = 0; i < NS; ++i) {
        const uint32_t q1 = (uint32_t)(q[i] + 1);
        top = q1 > top ? q1 : top;
        units += khg_hot_unit(q1);
        seen |= khg_hot_seen(q1, stream[i]);
    }
    if (top <= 2u && (seen & ~hot) == 0) {  // all of them -1, 0 or +1 and seen in their streams before
        hot += units;
    } else {
        KHG_UNROLL
        for (int i = 0; i < NS; ++i)
            q[i] = khg_count_rare<LHIST>(a, pl, lcnt, lfirst, hot, q[i], stream[i], (so + boff[i]) >> 1);
    }
    KHG_UNROLL
    for (int i = 0; i < NS; ++i) *(int16_t*)((char*)pl.sym + (so + boff[i])) = (int16_t)q[i];
}

// Finished, unscaled outputs (s2, d2) of row pair k of the lane's low-half column (L) and high-half column (H):
// scale, store / quantise.  rd: this position's reciprocal divisors per band slot; so, llo: byte offsets of
// position (k, pair) in the symbol plane and in the LL buffer.
template <int LEVEL, int LHIST>
KHG_DEV void khg_col_emit(const KhgFwdArgs& a, const KhgFwdPlane& pl, uint32_t* lcnt, uint32_t* lfirst, KhgHot& hot,
                          int pair, int k, double s2L, double d2L, double s2H, double d2H, const double* rd, uint32_t so,
                          uint32_t llo)
{
    const int hh = pl.ph / 2, hw = pl.pw / 2;
    const double loL = khg_fwd_scale_even(s2L), hiL = khg_fwd_scale_odd(d2L);
    const double loH = khg_fwd_scale_even(s2H), hiH = khg_fwd_scale_odd(d2H);
    if (pl.coef) {
        const uint32_t c0 = (uint32_t)k * (uint32_t)pl.coef_pitch + (uint32_t)pair;
        const uint32_t c1 = (uint32_t)(hh + k) * (uint32_t)pl.coef_pitch + (uint32_t)pair;
        pl.coef[c0] = loL;
        pl.coef[c0 + (uint32_t)hw] = loH;
        pl.coef[c1] = hiL;
        pl.coef[c1 + (uint32_t)hw] = hiH;
    }
    const bool emit = k >= pl.emit_k0 && k < pl.emit_k1;
    const uint32_t right = 2u * (uint32_t)hw, down = 2u * (uint32_t)hh * (uint32_t)pl.sym_pitch;
    if (LEVEL == 1) {  // bands LH(4) HL(5) HH(6) are final, LL feeds level 2 (Compress.h:154-160)
        *(double*)((char*)pl.ll + llo) = loL;
        if (emit) {
            const double c[3] = {loH, hiL, hiH};
            const int band[3] = {4, 5, 6}, stream[3] = {0, 1, 1};
            const int y[3] = {k, hh + k, hh + k}, x[3] = {hw + pair, pair, hw + pair};
            const uint32_t boff[3] = {right, down, down + right};
            khg_quant_emit<3, LHIST>(a, pl, lcnt, lfirst, hot, c, rd, band, stream, y, x, so, boff);
        }
    } else if (emit) {  // bands 0..3
        const double c[4] = {loL, loH, hiL, hiH};
        const int band[4] = {0, 1, 2, 3}, stream[4] = {0, 0, 1, 1};
        const int y[4] = {k, k, hh + k, hh + k}, x[4] = {pair, hw + pair, pair, hw + pair};
        const uint32_t boff[4] = {0, right, down, down + right};
        khg_quant_emit<4, LHIST>(a, pl, lcnt, lfirst, hot, c, rd, band, stream, y, x, so, boff);
    }
}

// One workgroup: KHG_WG_STRIPS adjacent strips (one per wave) over one segment of rows of NP planes of equal
// geometry (NP = 2: Cb and Cr from the same pixels).  wgl = workgroup index within the plane's grid.
template <int SRC, int LEVEL, int NP>
KHG_DEV void khg_fwd_wg_body(const KhgFwdArgs& a, const KhgFwdPlane* pls, int wgl, KhgFwdSmem<LEVEL, NP>& sh)
{
    typedef KhgFwdSmem<LEVEL, NP> SM;
    const KhgFwdPlane& g = pls[0];  // geometry (shared by the NP planes)
    const int swg = wgl % g.nstrip_wg, seg = wgl / g.nstrip_wg;
    const int np = g.pw / 2, hh = g.ph / 2, hw = g.pw / 2;
    int r0, r1;
    khg_segment_rows(g.row_begin, g.row_end, g.nseg, seg, &r0, &r1);
    const int k0 = r0 / 2, k1 = r1 / 2;
    const int P0 = KHG_STRIP_PAIRS * KHG_WG_STRIPS * swg;  // first output pair of the workgroup

    KhgLift colL[NP][KHG_NSLOT(256)], colH[NP][KHG_NSLOT(256)];
    KhgHot hots[NP][KHG_NSLOT(256)];
    KhgFwdRaw<SRC> raws[KHG_NSLOT(256)];
    KhgLaneSrc lsrc[KHG_NSLOT(256)];
    int pairs[KHG_NSLOT(256)];     // the lane's pair, clamped to the strip's output pairs
    bool outl[KHG_NSLOT(256)];     // the lane outputs its pair
    uint32_t symo[KHG_NSLOT(256)], llo[KHG_NSLOT(256)];  // byte offsets of (k, pair) in the symbol plane / LL buffer
    // values of the row pass that cross an exchange: [row a / b][plane]
    double E[2][NP][KHG_NSLOT(256)], O[2][NP][KHG_NSLOT(256)], D1[2][NP][KHG_NSLOT(256)], S1[2][NP][KHG_NSLOT(256)],
        D2[2][NP][KHG_NSLOT(256)], X[2][NP][KHG_NSLOT(256)];

    KHG_FOR_THREADS(256)
    {
        if (a.prof && tid == 0) khg_atomic_max64(&a.prof->prof_not_start, ~khg_clock());
        const int lane = tid & 63, s = swg * KHG_WG_STRIPS + KHG_UNIFORM(tid >> 6);
        // strip s outputs pairs 60 s .. 60 s + 59 (those inside the plane); its lanes hold pairs 60 s - 2 .. 60 s + 61
        const int v0 = KHG_STRIP_PAIRS * s, v1 = v0 + KHG_STRIP_PAIRS < np ? v0 + KHG_STRIP_PAIRS : np;
        const int p = v0 - 2 + lane;
        outl[KHG_SLOT] = p >= v0 && p < v1;  // a wave beyond the last strip loads and computes without output
        pairs[KHG_SLOT] = p < v0 ? v0 : (p > v0 + KHG_STRIP_PAIRS - 1 ? v0 + KHG_STRIP_PAIRS - 1 : p);
        lsrc[KHG_SLOT] = khg_fwd_lane_src<SRC>(p, g.pw);
        // the first trip finishes row pair k0 - 4 (nothing is stored before k0; the offsets just run along)
        symo[KHG_SLOT] = 2u * ((uint32_t)(k0 - 4) * (uint32_t)g.sym_pitch + (uint32_t)pairs[KHG_SLOT]);
        llo[KHG_SLOT] = 8u * ((uint32_t)(k0 - 4) * (uint32_t)g.ll_pitch + (uint32_t)pairs[KHG_SLOT]);
        const KhgLift z = {0, 0, 0, 0, 0};
        for (int n = 0; n < NP; ++n) {
            colL[n][KHG_SLOT] = z;
            colH[n][KHG_SLOT] = z;
            hots[n][KHG_SLOT] = 0;
        }
        // rows of the first trip
        uint32_t foa, fob;
        khg_fwd_row_offsets<SRC>(a, g, k0 - 2, foa, fob);
        khg_fwd_fetch<SRC>(a, g, raws[KHG_SLOT], foa, fob, lsrc[KHG_SLOT].col);
        const int lh = LEVEL == 1 ? (int)SM::LHIST : a.lhist, l0 = SM::LHIST_BIAS - lh / 2;  // the bins in use
        for (int n = 0; n < NP; ++n)
            for (int i = l0 + tid; i < l0 + lh; i += 256) {
                sh.cnt[n][i] = 0;
                sh.first[n][i] = 0xFFFFFFFFu;
            }
        // The divisor depends on the band and on x + y only (Compress.h:215-218); pair P0 + i at row pair k sits
        // on diagonal u = i + (k - k0) of each band this workgroup emits.
        const int nu = KHG_WG_STRIPS * KHG_STRIP_PAIRS + (k1 - k0);
        for (int e = tid; e < SM::NBAND * nu; e += 256) {
            int s2 = e / nu, u = e - s2 * nu;
            int band = LEVEL == 1 ? 4 + s2 : s2;
            int by = (LEVEL == 1 ? (band >= 5) : (band >= 2)) ? hh : 0;
            int bx = (LEVEL == 1 ? (band != 5) : (band & 1)) ? hw : 0;
            sh.rdiv[s2 * KHG_RD_N + KHG_RD_PAD + u] = 1.0 / khg_divisor(g.q, band, bx + P0 + u, by + k0, a.lvl);
        }
    }
    KHG_BARRIER();

    // Row pairs k0-2 .. k1+1 are fed; the column pipeline hands back pair j-2 when pair j goes in.
    for (int j = k0 - 2; j <= k1 + 1; ++j) {
        // source rows of the next trip (the same for every thread)
        uint32_t noa, nob;
        khg_fwd_row_offsets<SRC>(a, g, j + 1, noa, nob);
        KHG_FOR_THREADS(256)
        {
            KHG_UNROLL
            for (int rr = 0; rr < 2; ++rr)
                KHG_UNROLL
                for (int n = 0; n < NP; ++n)
                    khg_fwd_samples<SRC>(raws[KHG_SLOT], lsrc[KHG_SLOT], rr, n, E[rr][n][KHG_SLOT], O[rr][n][KHG_SLOT]);
            // next trip's rows: in flight during this trip's arithmetic (unconditional, also after the last
            // trip, where the rows are reflected back into the plane and the loads are simply not used: a
            // condition here would make the loads land in other registers and cost a wait and copies per trip)
            khg_fwd_fetch<SRC>(a, g, raws[KHG_SLOT], noa, nob, lsrc[KHG_SLOT].col);
        }
        // predict 1: d1[p] = o[p] + P1 * (e[p] + e[p+1])
        for (int rr = 0; rr < 2; ++rr)
            for (int n = 0; n < NP; ++n) KHG_LANE_NEXT(X[rr][n], E[rr][n])
        KHG_FOR_THREADS(256)
        {
            KHG_UNROLL
            for (int rr = 0; rr < 2; ++rr)
                KHG_UNROLL
                for (int n = 0; n < NP; ++n)
                    D1[rr][n][KHG_SLOT] = O[rr][n][KHG_SLOT] + KHG_P1 * (E[rr][n][KHG_SLOT] + X[rr][n][KHG_SLOT]);
        }
        // update 1: s1[p] = e[p] + U1 * (d1[p-1] + d1[p])
        for (int rr = 0; rr < 2; ++rr)
            for (int n = 0; n < NP; ++n) KHG_LANE_PREV(X[rr][n], D1[rr][n])
        KHG_FOR_THREADS(256)
        {
            KHG_UNROLL
            for (int rr = 0; rr < 2; ++rr)
                KHG_UNROLL
                for (int n = 0; n < NP; ++n)
                    S1[rr][n][KHG_SLOT] = E[rr][n][KHG_SLOT] + KHG_U1 * (X[rr][n][KHG_SLOT] + D1[rr][n][KHG_SLOT]);
        }
        // predict 2: d2[p] = d1[p] + P2 * (s1[p] + s1[p+1])
        for (int rr = 0; rr < 2; ++rr)
            for (int n = 0; n < NP; ++n) KHG_LANE_NEXT(X[rr][n], S1[rr][n])
        KHG_FOR_THREADS(256)
        {
            KHG_UNROLL
            for (int rr = 0; rr < 2; ++rr)
                KHG_UNROLL
                for (int n = 0; n < NP; ++n)
                    D2[rr][n][KHG_SLOT] = D1[rr][n][KHG_SLOT] + KHG_P2 * (S1[rr][n][KHG_SLOT] + X[rr][n][KHG_SLOT]);
        }
        // update 2: s2[p] = s1[p] + U2 * (d2[p-1] + d2[p]); scale; columns
        for (int rr = 0; rr < 2; ++rr)
            for (int n = 0; n < NP; ++n) KHG_LANE_PREV(X[rr][n], D2[rr][n])
        KHG_FOR_THREADS(256)
        {
            const int k = j - 2;
            const bool out = outl[KHG_SLOT] && k >= k0 && k < k1;
            const int pair = pairs[KHG_SLOT];
            const double* rd = &sh.rdiv[KHG_RD_PAD + (pair - P0) + (k - k0)];
            double rdv[SM::NBAND];
            KHG_UNROLL
            for (int s = 0; s < SM::NBAND; ++s) rdv[s] = rd[s * KHG_RD_N];
            KHG_UNROLL
            for (int n = 0; n < NP; ++n) {
                // row a is the even sample of both columns, row b the odd one
                double lo[2], hi[2];
                KHG_UNROLL
                for (int rr = 0; rr < 2; ++rr) {
                    const double s2 = S1[rr][n][KHG_SLOT] + KHG_U2 * (X[rr][n][KHG_SLOT] + D2[rr][n][KHG_SLOT]);
                    lo[rr] = khg_fwd_scale_even(s2);
                    hi[rr] = khg_fwd_scale_odd(D2[rr][n][KHG_SLOT]);
                }
                double s2L, d2L, s2H, d2H;
                khg_fwd_step(colL[n][KHG_SLOT], lo[0], lo[1], s2L, d2L);
                khg_fwd_step(colH[n][KHG_SLOT], hi[0], hi[1], s2H, d2H);
                if (out)
                    khg_col_emit<LEVEL, SM::LHIST>(a, pls[n], sh.cnt[n], sh.first[n], hots[n][KHG_SLOT], pair, k, s2L, d2L,
                                                   s2H, d2H, rdv, symo[KHG_SLOT], llo[KHG_SLOT]);
            }
            symo[KHG_SLOT] += 2u * (uint32_t)g.sym_pitch;
            llo[KHG_SLOT] += 8u * (uint32_t)g.ll_pitch;
        }
    }

    KHG_FOR_THREADS(256)
    {
        for (int n = 0; n < NP; ++n) {
            const KhgHot hot = hots[n][KHG_SLOT];
            for (int v = 0; v < 3; ++v) {  // byte v counts symbol 1 - v
                uint32_t c = (hot >> (8 * v)) & 0xFFu;
                if (c) khg_lds_add(&sh.cnt[n][SM::LHIST_BIAS + 1 - v], c);
            }
        }
    }
    KHG_BARRIER();
    KHG_FOR_THREADS(256)
    {
        const int lh = LEVEL == 1 ? (int)SM::LHIST : a.lhist, l0 = SM::LHIST_BIAS - lh / 2;
        for (int n = 0; n < NP; ++n)
            for (int i = l0 + tid; i < l0 + lh; i += 256) {
                uint32_t c = sh.cnt[n][i];
                if (c) {
                    int bin = i - SM::LHIST_BIAS + KHG_SYM_BIAS;
                    khg_atomic_add(&pls[n].hist_cnt[bin], c);
                    khg_atomic_min(&pls[n].hist_first[bin], sh.first[n][i]);
                }
            }
        if (a.prof && tid == 0) khg_atomic_max64(&a.prof->prof_end, khg_clock());
    }
}

// Launch kinds.
enum { KHG_FWD_RGB = 0, KHG_FWD_PLANES1 = 1, KHG_FWD_PLANES2 = 2, KHG_FWD_BGR = 3 };

#if defined(KHG_TIMING) && !defined(KHG_SIM)
// developer build: per-workgroup start / end stamps (100 MHz) and hardware id of the last forward launch
__device__ uint32_t khg_dbg_fwd[4 * 8192];
#define KHG_FWD_STAMP(wg, slot, v) \
    if (threadIdx.x == 0 && (wg) < 8192) khg_dbg_fwd[4 * (wg) + (slot)] = (v)
#else
#define KHG_FWD_STAMP(wg, slot, v)
#endif

// Workgroup `wg` of a launch: which plane it belongs to, and its body.
template <int KIND>
KHG_DEV void khg_fwd_launch_body(const KhgFwdArgs& a, int wg, KhgFwdSmemAny& sh)
{
    KHG_FWD_STAMP(wg, 0, (uint32_t)wall_clock64());
    KHG_FWD_STAMP(wg, 2, __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)));  // HW_ID
    KHG_FWD_STAMP(wg, 3, __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)));  // XCC_ID
    if constexpr (KIND == KHG_FWD_RGB) {
        if (a.nplanes < 2 || wg < a.pl[1].wg0) khg_fwd_wg_body<KHG_SRC_Y, 1, 1>(a, &a.pl[0], wg, sh.y);
        else khg_fwd_wg_body<KHG_SRC_C, 1, 2>(a, &a.pl[1], wg - a.pl[1].wg0, sh.c);
    } else if constexpr (KIND == KHG_FWD_BGR) {
        if (a.nplanes < 2 || wg < a.pl[1].wg0) khg_fwd_wg_body<KHG_SRC_Y_BGR, 1, 1>(a, &a.pl[0], wg, sh.y);
        else khg_fwd_wg_body<KHG_SRC_C_BGR, 1, 2>(a, &a.pl[1], wg - a.pl[1].wg0, sh.c);
    } else {
        int p = 0;
        while (p + 1 < a.nplanes && wg >= a.pl[p + 1].wg0) ++p;
        if constexpr (KIND == KHG_FWD_PLANES1) khg_fwd_wg_body<KHG_SRC_PLANE, 1, 1>(a, &a.pl[p], wg - a.pl[p].wg0, sh.y);
        else khg_fwd_wg_body<KHG_SRC_PLANE, 2, 1>(a, &a.pl[p], wg - a.pl[p].wg0, sh.l2);
    }
    KHG_FWD_STAMP(wg, 1, (uint32_t)wall_clock64());
}
