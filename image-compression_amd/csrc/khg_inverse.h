// khg_inverse.h -- dequantiser + one inverse 2-D CDF 9/7 lifting level, and the final
// YCbCr -> RGB conversion.
//
// Mirror image of khg_forward.h, in the same lane-streaming layout (IWT2D runs columns first, then rows:
// DWT.h:143-165).  A wave owns a strip of 64 pairs of output columns and walks down a segment of rows, one packed
// row pair (low-pass row k, high-pass row hh+k) per trip; lane l holds pair p = first pair + l:
//   columns: the lane's two packed source columns (low half p, high half hw+p) are read -- symbols dequantised on
//            the fly (Decompress.h:98-123), the level-1 LL quadrant as f64 -- and fed to two streaming inverse
//            lifting pipelines held in registers, which hand back two finished natural-order rows;
//   rows   : each of the two rows is inverse-lifted across the lanes, the neighbouring pair's value coming from
//            the neighbouring lane by a DPP wave shift (four exchanges per row, two halo pairs per strip side), and
//            the lane's two samples are stored -- or, in the fused level-1 launch, combined with the finished
//            chroma sample of this position (one Cb, one Cr per lane and trip: Colorspace.h:31-33 reads them at
//            [y/2][x/2]) into RGB and written as u8, so the f64 planes never exist in memory.
#pragma once
#include "khg_common.h"
#include "khg_decode.h"
#include "khg_forward.h"

enum {
    KHG_DV_PAD = 4,
    KHG_DV_N = KHG_WG_STRIPS * KHG_STRIP_PAIRS + 4 + KHG_MAX_SEG_ROWS / 2 + 2 * KHG_DV_PAD + 8  // divisors per band
};

struct KhgInvPlane {
    const int16_t* sym;   // symbol plane of this colour component
    int sym_pitch;
    const double* ll;     // level 1: the LL quadrant (output of the level-2 inverse)
    int ll_pitch;
    int ph, pw;           // rows, cols of this transform
    double* dst;          // natural-order output, ph x pw (unused by the fused launch)
    int dst_pitch;
    int row_begin, row_end;  // output rows this launch produces (even; the whole plane, or one tile's band)
    int seg_rows;         // output rows of the tallest workgroup segment (even; see khg_segment_rows)
    int nseg;
    int nstrip_wg;        // workgroups side by side (KHG_WG_STRIPS strips each)
    int wg0;              // first workgroup of this plane in the launch
    KhgQuant q[3];        // quantiser constants of this plane type for presets 0..2
};

struct KhgInvArgs {
    const KhgDecHeader* hdr;  // supplies the preset (lvl) and the go/no-go flag
    int nplanes;
    int nwg;
    int out_row0;             // fused launch: image row held by the first row of r/g/b (0 for a whole image)
    uint8_t *r, *g, *b;       // fused launch: the output image, pitch pw
    // BMP variant: the output is a 24-bit BMP pixel array instead (KhuGleSignal.cpp:285-293: B, G, R per pixel, image
    // row y at row ph - 1 - y of the array); r = its first byte, bgr_stride = bytes per row of the array
    size_t bgr_stride;
    KhgInvPlane pl[3];
};

template <int LEVEL>
struct KhgInvSmem {
    enum { NBAND = LEVEL == 1 ? 3 : 4 };
    double dv[NBAND * KHG_DV_N];  // divisor of band slot s at diagonal u: [s * KHG_DV_N + u]
};
struct KhgInvFusedSmem;

// Streaming state of one 1-D inverse lifting line after feeding pair j: d[j], s1[j], d1[j-1], s0[j-1].
struct KhgILift {
    double d, s1, d1, s0;
};

// With s[k] = low[k]/A, d[k] = high[k]*A (DWT.h:76-82, A = 1.149604398) the steps of DWT.h:84-114 read
//   s1[k] = s[k]  + C1*(d[k-1]  + d[k])        undo update 2
//   d1[k] = d[k]  + C2*(s1[k]   + s1[k+1])     undo predict 2
//   s0[k] = s1[k] + C3*(d1[k-1] + d1[k])       undo update 1
//   d0[k] = d1[k] + C4*(s0[k]   + s0[k+1])     undo predict 1
#define KHG_IA (1.149604398)
#define KHG_C1 (-0.4435068522)
#define KHG_C2 (-0.8829110762)
#define KHG_C3 (0.05298011854)
#define KHG_C4 (1.586134342)

KHG_DEV double khg_inv_unscale_low(double lo) { const double rA = 1 / KHG_IA; return khg_div_const(lo, KHG_IA, rA); }
KHG_DEV double khg_inv_unscale_high(double hi) { return hi * KHG_IA; }

// Feed packed pair j = (low[j], high[j]); returns natural samples x[2(j-2)], x[2(j-2)+1].
KHG_DEV void khg_inv_step(KhgILift& st, double lo, double hi, double& xe, double& xo)
{
    double sn = khg_inv_unscale_low(lo), dn = khg_inv_unscale_high(hi);
    double s1n = sn + KHG_C1 * (st.d + dn);
    double d1n = st.d + KHG_C2 * (st.s1 + s1n);
    double s0n = st.s1 + KHG_C3 * (st.d1 + d1n);
    double d0 = st.d1 + KHG_C4 * (st.s0 + s0n);
    xe = st.s0;
    xo = d0;
    st.d = dn;
    st.s1 = s1n;
    st.d1 = d1n;
    st.s0 = s0n;
}

// Raw inputs of one lane for one trip: the symbols at (ke, p), (ke, hw+p), (ko, p), (ko, hw+p) -- at level 1 the
// first one is the LL sample instead.
struct KhgInvRaw {
    int q[4];
    double ll;
};

KHG_DEV uint32_t khg_clamp_u8(double v)
{
    // Main.cpp:307-315: max(0, min(v, 255)) with the Win32 macros (a<b?a:b), (a>b?a:b), then truncation to
    // unsigned char.  v is never NaN here (finite symbols times finite constants); the conversion to an unsigned
    // integer truncates toward zero and saturates at zero, which is what clamping below does to every v < 1, and
    // the integer minimum with 255 is what clamping above does to every v >= 255.
#ifdef KHG_SIM
    const double c = v < 0.0 ? 0.0 : (v > 4e9 ? 4e9 : v);
    const uint32_t u = (uint32_t)c;
#else
    const uint32_t u = (uint32_t)v;  // v_cvt_u32_f64 saturates
#endif
    return u < 255u ? u : 255u;
}

// Colorspace.h:25-36 + Main.cpp:307-315 for the two pixels that share one chroma site: products and sums
// left to right, clamp, truncate.  Returns the pixels' bytes in bits 0..7 and 8..15 of rgb[0..2].
KHG_DEV void khg_rgb_pair(double y0, double y1, double Cb, double Cr, uint32_t rgb[3])
{
    const double rr = Cr * 1.402, gb = Cb * -0.34414, gr = Cr * -0.71414, bb = Cb * 1.772;
    rgb[0] = khg_clamp_u8(y0 * 1 + rr) | (khg_clamp_u8(y1 * 1 + rr) << 8);
    rgb[1] = khg_clamp_u8(y0 * 1 + gb + gr) | (khg_clamp_u8(y1 * 1 + gb + gr) << 8);
    rgb[2] = khg_clamp_u8(y0 * 1 + bb) | (khg_clamp_u8(y1 * 1 + bb) << 8);
}

// Where the fused level-1 launch puts its pixels: the planar u8 image or a BMP pixel array.
enum { KHG_OUT_RGB = 1, KHG_OUT_BGR = 2 };

KHG_DEV void khg_st16(uint8_t* p, uint32_t v) { const uint16_t h = (uint16_t)v; __builtin_memcpy(p, &h, 2); }

// One workgroup: KHG_WG_STRIPS adjacent strips (one per wave) over one segment of output rows of plane pl, into
// its f64 output.
template <int LEVEL>
KHG_DEV void khg_inv_wg_body(const KhgInvArgs& a, const KhgInvPlane& pl, int wgl, KhgInvSmem<LEVEL>& sh)
{
    typedef KhgInvSmem<LEVEL> SM;
    if (!a.hdr->ok) return;
    const int lvl = a.hdr->lvl;
    const KhgQuant& qc = pl.q[lvl];
    const int swg = wgl % pl.nstrip_wg, seg = wgl / pl.nstrip_wg;
    const int np = pl.pw / 2, hh = pl.ph / 2, hw = pl.pw / 2;
    int r0, r1;
    khg_segment_rows(pl.row_begin, pl.row_end, pl.nseg, seg, &r0, &r1);
    const int k0 = r0 / 2, k1 = r1 / 2;
    const int P0 = KHG_STRIP_PAIRS * KHG_WG_STRIPS * swg - 2;  // pair held by the workgroup's first lane
    // packed rows this segment reads (reflections included) lie in [klo, klo + nk)
    const int klo = k0 - 4 < 0 ? 0 : k0 - 4;
    const int khi = k1 + 3 > hh - 1 ? hh - 1 : k1 + 3;

    KhgILift colL[KHG_NSLOT(256)], colH[KHG_NSLOT(256)];
    KhgInvRaw raws[KHG_NSLOT(256)];
    int pairs[KHG_NSLOT(256)];  // the lane's pair, clamped to the strip's output pairs
    // Columns the lane's low-pass and high-pass samples are read from.  Lanes hold pairs 60 s - 2 .. 60 s + 61 of the
    // interleaved row; a pair outside the plane is the whole-sample mirror image (x[-i] = x[i], x[n-1+i] =
    // x[n-1-i]) of samples inside, i.e. a low-pass sample of one column and a high-pass sample of another.  Lifting
    // preserves the symmetry, so the interior formula gives what the reference's edge lines give (DWT.h:75-100)
    // and the row pass needs no edge case.
    int colsL[KHG_NSLOT(256)], colsH[KHG_NSLOT(256)];
    int uL[KHG_NSLOT(256)], uH[KHG_NSLOT(256)];  // their offsets in the workgroup's divisor diagonals
    bool outl[KHG_NSLOT(256)];  // the lane outputs its pair
    double LO[2][KHG_NSLOT(256)], HI[2][KHG_NSLOT(256)], S1[2][KHG_NSLOT(256)], D1[2][KHG_NSLOT(256)],
        S0[2][KHG_NSLOT(256)], X[2][KHG_NSLOT(256)];

    // Packed row indices of trip j: low-pass row ke, high-pass row ko (whole-sample reflection of the natural rows
    // 2j and 2j+1 at the plane's top and bottom).
#define KHG_INV_KE(j) (khg_reflect(2 * (j), pl.ph) >> 1)
#define KHG_INV_KO(j) (khg_reflect(2 * (j) + 1, pl.ph) >> 1)

    KHG_FOR_THREADS(256)
    {
        const int lane = tid & 63, s = swg * KHG_WG_STRIPS + KHG_UNIFORM(tid >> 6);
        const int v0 = KHG_STRIP_PAIRS * s, v1 = v0 + KHG_STRIP_PAIRS < np ? v0 + KHG_STRIP_PAIRS : np;
        const int p = v0 - 2 + lane;
        outl[KHG_SLOT] = p >= v0 && p < v1;  // a wave beyond the last strip loads and computes without output
        pairs[KHG_SLOT] = p < v0 ? v0 : (p > v0 + KHG_STRIP_PAIRS - 1 ? v0 + KHG_STRIP_PAIRS - 1 : p);
        if (pairs[KHG_SLOT] > np - 1) pairs[KHG_SLOT] = np - 1;
        colsL[KHG_SLOT] = khg_reflect(2 * p, pl.pw) >> 1;
        colsH[KHG_SLOT] = khg_reflect(2 * p + 1, pl.pw) >> 1;
        const int umax = KHG_WG_STRIPS * KHG_STRIP_PAIRS + 3;  // lanes beyond the plane: any entry of the table
        int u = colsL[KHG_SLOT] - P0;
        uL[KHG_SLOT] = u < 0 ? 0 : (u > umax ? umax : u);
        u = colsH[KHG_SLOT] - P0;
        uH[KHG_SLOT] = u < 0 ? 0 : (u > umax ? umax : u);
        const KhgILift z = {0, 0, 0, 0};
        colL[KHG_SLOT] = z;
        colH[KHG_SLOT] = z;
        // Decompress.h:105, 120-121: coefficient = symbol * divisor of its band and position; the divisor depends on
        // the band and on x + y only: pair P0 + i at packed row klo + r sits on diagonal u = i + r.
        const int nu = KHG_WG_STRIPS * KHG_STRIP_PAIRS + 4 + (khi - klo + 1);
        KHG_UNROLL
        for (int s2 = 0; s2 < SM::NBAND; ++s2) {
            const int band = LEVEL == 1 ? 4 + s2 : s2;
            const int by = (LEVEL == 1 ? (band >= 5) : (band >= 2)) ? hh : 0;
            const int bx = (LEVEL == 1 ? (band != 5) : (band & 1)) ? hw : 0;
            for (int u = tid; u < nu; u += 256) sh.dv[s2 * KHG_DV_N + u] = khg_divisor(qc, band, bx + P0 + u, by + klo, lvl);
        }
    }
    KHG_BARRIER();

    // Loads of trip j into raw (all four positions).
#define KHG_INV_FETCH(raw, j)                                                                               \
    {                                                                                                       \
        const int ke_ = KHG_INV_KE(j), ko_ = hh + KHG_INV_KO(j);                                            \
        const uint32_t oe_ = (uint32_t)ke_ * (uint32_t)pl.sym_pitch, oo_ = (uint32_t)ko_ * (uint32_t)pl.sym_pitch; \
        const uint32_t cl_ = (uint32_t)colsL[KHG_SLOT], ch_ = (uint32_t)(hw + colsH[KHG_SLOT]);             \
        if (LEVEL == 1) {                                                                                   \
            (raw).ll = pl.ll[(uint32_t)ke_ * (uint32_t)pl.ll_pitch + cl_];                                  \
            (raw).q[0] = 0;                                                                                 \
        } else {                                                                                            \
            (raw).ll = 0.0;                                                                                 \
            (raw).q[0] = pl.sym[oe_ + cl_];                                                                 \
        }                                                                                                   \
        (raw).q[1] = pl.sym[oe_ + ch_];                                                                     \
        (raw).q[2] = pl.sym[oo_ + cl_];                                                                     \
        (raw).q[3] = pl.sym[oo_ + ch_];                                                                     \
    }

    KHG_FOR_THREADS(256) { KHG_INV_FETCH(raws[KHG_SLOT], k0 - 2) }

    for (int j = k0 - 2; j <= k1 + 1; ++j) {
        KHG_FOR_THREADS(256)
        {
            // ---- column phase: dequantise, feed the two column pipelines, get natural rows 2(j-2), 2(j-2)+1
            const KhgInvRaw raw = raws[KHG_SLOT];
            const int ke = KHG_INV_KE(j), ko = KHG_INV_KO(j);
            const int re = ke - klo, ro = ko - klo;
            double c[4];
            if (LEVEL == 1) {
                c[0] = raw.ll;
                c[1] = (double)raw.q[1] * sh.dv[0 * KHG_DV_N + uH[KHG_SLOT] + re];  // band 4: (ke, hw + column)
                c[2] = (double)raw.q[2] * sh.dv[1 * KHG_DV_N + uL[KHG_SLOT] + ro];  // band 5: (hh + ko, column)
                c[3] = (double)raw.q[3] * sh.dv[2 * KHG_DV_N + uH[KHG_SLOT] + ro];  // band 6: (hh + ko, hw + column)
            } else {
                c[0] = (double)raw.q[0] * sh.dv[0 * KHG_DV_N + uL[KHG_SLOT] + re];
                c[1] = (double)raw.q[1] * sh.dv[1 * KHG_DV_N + uH[KHG_SLOT] + re];
                c[2] = (double)raw.q[2] * sh.dv[2 * KHG_DV_N + uL[KHG_SLOT] + ro];
                c[3] = (double)raw.q[3] * sh.dv[3 * KHG_DV_N + uH[KHG_SLOT] + ro];
            }
            // next trip's inputs: in flight during this trip's arithmetic (unconditional: after the last trip the
            // rows are reflected back into the plane and the loads are not used)
            KHG_INV_FETCH(raws[KHG_SLOT], j + 1)
            khg_inv_step(colL[KHG_SLOT], c[0], c[2], LO[0][KHG_SLOT], LO[1][KHG_SLOT]);
            khg_inv_step(colH[KHG_SLOT], c[1], c[3], HI[0][KHG_SLOT], HI[1][KHG_SLOT]);
            // ---- row phase on the two rows: s = low / A, d = high * A
            KHG_UNROLL
            for (int rr = 0; rr < 2; ++rr) {
                LO[rr][KHG_SLOT] = khg_inv_unscale_low(LO[rr][KHG_SLOT]);
                HI[rr][KHG_SLOT] = khg_inv_unscale_high(HI[rr][KHG_SLOT]);
            }
        }
        // s1[p] = s[p] + C1 * (d[p-1] + d[p])
        for (int rr = 0; rr < 2; ++rr) KHG_LANE_PREV(X[rr], HI[rr])
        KHG_FOR_THREADS(256)
        {
            KHG_UNROLL
            for (int rr = 0; rr < 2; ++rr) {
                S1[rr][KHG_SLOT] = LO[rr][KHG_SLOT] + KHG_C1 * (X[rr][KHG_SLOT] + HI[rr][KHG_SLOT]);
            }
        }
        // d1[p] = d[p] + C2 * (s1[p] + s1[p+1])
        for (int rr = 0; rr < 2; ++rr) KHG_LANE_NEXT(X[rr], S1[rr])
        KHG_FOR_THREADS(256)
        {
            KHG_UNROLL
            for (int rr = 0; rr < 2; ++rr) {
                D1[rr][KHG_SLOT] = HI[rr][KHG_SLOT] + KHG_C2 * (S1[rr][KHG_SLOT] + X[rr][KHG_SLOT]);
            }
        }
        // s0[p] = s1[p] + C3 * (d1[p-1] + d1[p])
        for (int rr = 0; rr < 2; ++rr) KHG_LANE_PREV(X[rr], D1[rr])
        KHG_FOR_THREADS(256)
        {
            KHG_UNROLL
            for (int rr = 0; rr < 2; ++rr) {
                S0[rr][KHG_SLOT] = S1[rr][KHG_SLOT] + KHG_C3 * (X[rr][KHG_SLOT] + D1[rr][KHG_SLOT]);
            }
        }
        // d0[p] = d1[p] + C4 * (s0[p] + s0[p+1]); x[2p] = s0[p], x[2p+1] = d0[p]
        for (int rr = 0; rr < 2; ++rr) KHG_LANE_NEXT(X[rr], S0[rr])
        KHG_FOR_THREADS(256)
        {
            const int m = j - 2;  // finished row pair
            if (outl[KHG_SLOT] && m >= k0 && m < k1) {
                const int pair = pairs[KHG_SLOT];
                double xe[2], xo[2];
                KHG_UNROLL
                for (int rr = 0; rr < 2; ++rr) {
                    xe[rr] = S0[rr][KHG_SLOT];
                    xo[rr] = D1[rr][KHG_SLOT] + KHG_C4 * (S0[rr][KHG_SLOT] + X[rr][KHG_SLOT]);
                }
                KHG_UNROLL
                for (int rr = 0; rr < 2; ++rr) {
                    double* out = pl.dst + ((uint32_t)(2 * m + rr) * (uint32_t)pl.dst_pitch + 2u * (uint32_t)pair);
                    const double v[2] = {xe[rr], xo[rr]};
                    __builtin_memcpy(out, v, 16);
                }
            }
        }
    }
#undef KHG_INV_FETCH
#undef KHG_INV_KE
#undef KHG_INV_KO
}

// ---- level 1 of all three components and the colour conversion in one launch.
//
// The launch above that turns the luma plane into RGB reads the finished Cb and Cr planes, which a launch before it
// wrote: 8 bytes per pixel go to memory and come back.  Here the workgroup that reconstructs a block of luma rows
// reconstructs the chroma samples under it itself: its four waves, which hold four adjacent luma strips (240
// pairs = 240 chroma columns), are also the four chroma strips of that block (two of Cb, two of Cr, 60 pairs = 120
// columns each).  A chroma trip finishes two chroma rows, which serve four luma rows: every other luma trip is
// preceded by one chroma trip, whose rows go through LDS (two buffers, one barrier per chroma trip) to the lanes
// that hold the luma pairs above them.  Same arithmetic, operation for operation, as the separate launches.

#define KHG_NS KHG_NSLOT(256)
struct KhgInvStrip {  // a thread's share of one strip's streaming state (level 1)
    KhgILift colL[KHG_NS], colH[KHG_NS];
    KhgInvRaw raw[KHG_NS];  // the next trip's inputs
    int pairs[KHG_NS], colsL[KHG_NS], colsH[KHG_NS], uL[KHG_NS], uH[KHG_NS];
    bool outl[KHG_NS];
    const int16_t* sym[KHG_NS];
    const double* ll[KHG_NS];
    double LO[2][KHG_NS], HI[2][KHG_NS], S1[2][KHG_NS], D1[2][KHG_NS], S0[2][KHG_NS], X[2][KHG_NS], XO[2][KHG_NS];
};

struct KhgInvGeo {  // the plane and the workgroup's place in it: the same for every thread
    int ph, pw, hh, hw, np, sym_pitch, ll_pitch, klo, P0;
    const double* dv;  // 3 bands x KHG_DV_N divisors
};

// A thread (lane `lane` of the wave that holds strip s) takes its place; inside KHG_FOR_THREADS.
KHG_DEV void khg_istrip_init(KhgInvStrip& st, int slot, const KhgInvGeo& g, int s, int lane, int ustrips,
                             const int16_t* sym, const double* ll)
{
    const int v0 = KHG_STRIP_PAIRS * s, v1 = v0 + KHG_STRIP_PAIRS < g.np ? v0 + KHG_STRIP_PAIRS : g.np;
    const int p = v0 - 2 + lane;
    st.outl[slot] = p >= v0 && p < v1;
    int pr = p < v0 ? v0 : (p > v0 + KHG_STRIP_PAIRS - 1 ? v0 + KHG_STRIP_PAIRS - 1 : p);
    if (pr > g.np - 1) pr = g.np - 1;
    st.pairs[slot] = pr;
    st.colsL[slot] = khg_reflect(2 * p, g.pw) >> 1;
    st.colsH[slot] = khg_reflect(2 * p + 1, g.pw) >> 1;
    const int umax = ustrips * KHG_STRIP_PAIRS + 3;  // lanes beyond the plane: any entry of the table
    int u = st.colsL[slot] - g.P0;
    st.uL[slot] = u < 0 ? 0 : (u > umax ? umax : u);
    u = st.colsH[slot] - g.P0;
    st.uH[slot] = u < 0 ? 0 : (u > umax ? umax : u);
    const KhgILift z = {0, 0, 0, 0};
    st.colL[slot] = z;
    st.colH[slot] = z;
    st.sym[slot] = sym;
    st.ll[slot] = ll;
}

// Divisors of the three level-1 bands along the diagonals the workgroup touches (Decompress.h:105, 120-121): pair
// P0 + i at packed row klo + r sits on diagonal u = i + r.  Inside KHG_FOR_THREADS.
KHG_DEV void khg_idv_fill(double* dv, const KhgInvGeo& g, const KhgQuant& qc, int lvl, int nu, int tid)
{
    KHG_UNROLL
    for (int s2 = 0; s2 < 3; ++s2) {
        const int band = 4 + s2;
        const int by = band >= 5 ? g.hh : 0, bx = band != 5 ? g.hw : 0;
        for (int u = tid; u < nu; u += 256) dv[s2 * KHG_DV_N + u] = khg_divisor(qc, band, bx + g.P0 + u, by + g.klo, lvl);
    }
}

// Loads of trip j; inside KHG_FOR_THREADS.
KHG_DEV void khg_istrip_fetch(KhgInvStrip& st, int slot, const KhgInvGeo& g, int j)
{
    const int ke = khg_reflect(2 * j, g.ph) >> 1, ko = g.hh + (khg_reflect(2 * j + 1, g.ph) >> 1);
    const uint32_t oe = (uint32_t)ke * (uint32_t)g.sym_pitch, oo = (uint32_t)ko * (uint32_t)g.sym_pitch;
    const uint32_t cl = (uint32_t)st.colsL[slot], ch = (uint32_t)(g.hw + st.colsH[slot]);
    const int16_t* sym = st.sym[slot];
    KhgInvRaw r;
    r.ll = st.ll[slot][(uint32_t)ke * (uint32_t)g.ll_pitch + cl];
    r.q[0] = 0;
    r.q[1] = sym[oe + ch];
    r.q[2] = sym[oo + cl];
    r.q[3] = sym[oo + ch];
    st.raw[slot] = r;
}

// Trip j of a strip: afterwards S0[rr] / XO[rr] hold the lane's even / odd sample of the natural rows 2 (j - 2) + rr.
// A workgroup collective (not inside KHG_FOR_THREADS).
KHG_DEV void khg_istrip_trip(KhgInvStrip& st, const KhgInvGeo& g, int j)
{
    KHG_FOR_THREADS(256)
    {
        const KhgInvRaw raw = st.raw[KHG_SLOT];
        const int re = (khg_reflect(2 * j, g.ph) >> 1) - g.klo, ro = (khg_reflect(2 * j + 1, g.ph) >> 1) - g.klo;
        const double c0 = raw.ll;
        const double c1 = (double)raw.q[1] * g.dv[0 * KHG_DV_N + st.uH[KHG_SLOT] + re];  // band 4: (ke, hw + column)
        const double c2 = (double)raw.q[2] * g.dv[1 * KHG_DV_N + st.uL[KHG_SLOT] + ro];  // band 5: (hh + ko, column)
        const double c3 = (double)raw.q[3] * g.dv[2 * KHG_DV_N + st.uH[KHG_SLOT] + ro];  // band 6: (hh + ko, hw + column)
        khg_istrip_fetch(st, KHG_SLOT, g, j + 1);  // in flight during this trip's arithmetic
        khg_inv_step(st.colL[KHG_SLOT], c0, c2, st.LO[0][KHG_SLOT], st.LO[1][KHG_SLOT]);
        khg_inv_step(st.colH[KHG_SLOT], c1, c3, st.HI[0][KHG_SLOT], st.HI[1][KHG_SLOT]);
        KHG_UNROLL
        for (int rr = 0; rr < 2; ++rr) {
            st.LO[rr][KHG_SLOT] = khg_inv_unscale_low(st.LO[rr][KHG_SLOT]);
            st.HI[rr][KHG_SLOT] = khg_inv_unscale_high(st.HI[rr][KHG_SLOT]);
        }
    }
    for (int rr = 0; rr < 2; ++rr) KHG_LANE_PREV(st.X[rr], st.HI[rr])
    KHG_FOR_THREADS(256)
    {
        KHG_UNROLL
        for (int rr = 0; rr < 2; ++rr)
            st.S1[rr][KHG_SLOT] = st.LO[rr][KHG_SLOT] + KHG_C1 * (st.X[rr][KHG_SLOT] + st.HI[rr][KHG_SLOT]);
    }
    for (int rr = 0; rr < 2; ++rr) KHG_LANE_NEXT(st.X[rr], st.S1[rr])
    KHG_FOR_THREADS(256)
    {
        KHG_UNROLL
        for (int rr = 0; rr < 2; ++rr)
            st.D1[rr][KHG_SLOT] = st.HI[rr][KHG_SLOT] + KHG_C2 * (st.S1[rr][KHG_SLOT] + st.X[rr][KHG_SLOT]);
    }
    for (int rr = 0; rr < 2; ++rr) KHG_LANE_PREV(st.X[rr], st.D1[rr])
    KHG_FOR_THREADS(256)
    {
        KHG_UNROLL
        for (int rr = 0; rr < 2; ++rr)
            st.S0[rr][KHG_SLOT] = st.S1[rr][KHG_SLOT] + KHG_C3 * (st.X[rr][KHG_SLOT] + st.D1[rr][KHG_SLOT]);
    }
    for (int rr = 0; rr < 2; ++rr) KHG_LANE_NEXT(st.X[rr], st.S0[rr])
    KHG_FOR_THREADS(256)
    {
        KHG_UNROLL
        for (int rr = 0; rr < 2; ++rr)
            st.XO[rr][KHG_SLOT] = st.D1[rr][KHG_SLOT] + KHG_C4 * (st.S0[rr][KHG_SLOT] + st.X[rr][KHG_SLOT]);
    }
}

enum { KHG_FUSED_COLS = KHG_WG_STRIPS * KHG_STRIP_PAIRS };  // chroma columns under a workgroup's luma strips
struct KhgInvFusedSmem {
    double dv[3 * KHG_DV_N];    // the luma plane's divisors
    double dvc[3 * KHG_DV_N];   // the chroma planes' (Cb and Cr share their quantiser constants)
    double cx[2][2][2][KHG_FUSED_COLS];  // [buffer][Cb, Cr][row of the pair][column - the workgroup's first column]
};

// One workgroup of the fused launch: the segment of luma rows [r0, r1) of four adjacent strips of a.pl[0], from
// a.pl[0] (luma), a.pl[1] (Cb) and a.pl[2] (Cr), into the RGB image or the BMP pixel array.
template <int OUT>
KHG_DEV void khg_inv_fused_wg_body(const KhgInvArgs& a, int wgl, KhgInvFusedSmem& sh)
{
    if (!a.hdr->ok) return;
    const int lvl = a.hdr->lvl;
    const KhgInvPlane& pl = a.pl[0];
    const KhgInvPlane& pc = a.pl[1];
    const int swg = wgl % pl.nstrip_wg, seg = wgl / pl.nstrip_wg;
    int r0, r1;
    khg_segment_rows(pl.row_begin, pl.row_end, pl.nseg, seg, &r0, &r1);
    const int k0 = r0 / 2, k1 = r1 / 2;            // luma row pairs = chroma rows [k0, k1)
    const int kc0 = k0 >> 1, kc1 = (k1 + 1) >> 1;  // chroma row pairs
    KhgInvGeo gy, gc;
    gy.ph = pl.ph; gy.pw = pl.pw; gy.hh = pl.ph / 2; gy.hw = pl.pw / 2; gy.np = pl.pw / 2;
    gy.sym_pitch = pl.sym_pitch; gy.ll_pitch = pl.ll_pitch;
    gy.klo = k0 - 4 < 0 ? 0 : k0 - 4;
    gy.P0 = KHG_STRIP_PAIRS * KHG_WG_STRIPS * swg - 2;
    gy.dv = sh.dv;
    const int khi = k1 + 3 > gy.hh - 1 ? gy.hh - 1 : k1 + 3;
    gc.ph = pc.ph; gc.pw = pc.pw; gc.hh = pc.ph / 2; gc.hw = pc.pw / 2; gc.np = pc.pw / 2;
    gc.sym_pitch = pc.sym_pitch; gc.ll_pitch = pc.ll_pitch;
    gc.klo = kc0 - 4 < 0 ? 0 : kc0 - 4;
    gc.P0 = KHG_STRIP_PAIRS * (KHG_WG_STRIPS / 2) * swg - 2;
    gc.dv = sh.dvc;
    const int khic = kc1 + 3 > gc.hh - 1 ? gc.hh - 1 : kc1 + 3;
    const int col0 = KHG_FUSED_COLS * swg;  // the workgroup's first luma pair = its first chroma column

    KhgInvStrip sy, sc;
    int cplane[KHG_NS];  // 0: the thread's wave does a strip of Cb, 1: of Cr
    KHG_FOR_THREADS(256)
    {
        const int lane = tid & 63, w = KHG_UNIFORM(tid >> 6);
        khg_istrip_init(sy, KHG_SLOT, gy, swg * KHG_WG_STRIPS + w, lane, KHG_WG_STRIPS, pl.sym, pl.ll);
        const KhgInvPlane& mine = a.pl[1 + (w >> 1)];
        cplane[KHG_SLOT] = w >> 1;
        khg_istrip_init(sc, KHG_SLOT, gc, swg * (KHG_WG_STRIPS / 2) + (w & 1), lane, KHG_WG_STRIPS / 2, mine.sym, mine.ll);
        khg_idv_fill(sh.dv, gy, pl.q[lvl], lvl, KHG_WG_STRIPS * KHG_STRIP_PAIRS + 4 + (khi - gy.klo + 1), tid);
        khg_idv_fill(sh.dvc, gc, pc.q[lvl], lvl, (KHG_WG_STRIPS / 2) * KHG_STRIP_PAIRS + 4 + (khic - gc.klo + 1), tid);
    }
    KHG_BARRIER();
    KHG_FOR_THREADS(256)
    {
        khg_istrip_fetch(sy, KHG_SLOT, gy, k0 - 2);
        khg_istrip_fetch(sc, KHG_SLOT, gc, kc0 - 2);
    }
    // the trips that only fill the pipelines
    for (int jc = kc0 - 2; jc <= kc0 + 1; ++jc) khg_istrip_trip(sc, gc, jc);
    for (int j = k0 - 2; j <= k0 + 1; ++j) khg_istrip_trip(sy, gy, j);
    for (int kc = kc0; kc < kc1; ++kc) {
        khg_istrip_trip(sc, gc, kc + 2);  // chroma rows 2 kc and 2 kc + 1
        KHG_FOR_THREADS(256)
        {
            if (sc.outl[KHG_SLOT]) {
                const int lc = 2 * sc.pairs[KHG_SLOT] - col0;
                KHG_UNROLL
                for (int rr = 0; rr < 2; ++rr) {
                    const double v[2] = {sc.S0[rr][KHG_SLOT], sc.XO[rr][KHG_SLOT]};
                    __builtin_memcpy(&sh.cx[kc & 1][cplane[KHG_SLOT]][rr][lc], v, 16);
                }
            }
        }
        KHG_BARRIER();
        const int m0 = 2 * kc < k0 ? k0 : 2 * kc, m1 = 2 * kc + 2 > k1 ? k1 : 2 * kc + 2;
        for (int m = m0; m < m1; ++m) {
            khg_istrip_trip(sy, gy, m + 2);  // luma rows 2 m and 2 m + 1, chroma row m
            KHG_FOR_THREADS(256)
            {
                if (sy.outl[KHG_SLOT]) {
                    const int pair = sy.pairs[KHG_SLOT];
                    const double Cb = sh.cx[kc & 1][0][m & 1][pair - col0], Cr = sh.cx[kc & 1][1][m & 1][pair - col0];
                    KHG_UNROLL
                    for (int rr = 0; rr < 2; ++rr) {
                        uint32_t rgb[3];
                        khg_rgb_pair(sy.S0[rr][KHG_SLOT], sy.XO[rr][KHG_SLOT], Cb, Cr, rgb);
                        if (OUT == KHG_OUT_BGR) {  // B0 G0 R0 B1 G1 R1, rows bottom-up
                            uint8_t* px = a.r + (size_t)(pl.ph - 1 - (2 * m + rr)) * a.bgr_stride + 6u * (uint32_t)pair;
                            khg_st16(px, (rgb[2] & 255u) | ((rgb[1] & 255u) << 8));
                            khg_st16(px + 2, (rgb[0] & 255u) | (rgb[2] & 0xFF00u));
                            khg_st16(px + 4, (rgb[1] >> 8) | (rgb[0] & 0xFF00u));
                        } else {
                            const uint32_t o = (uint32_t)(2 * m + rr - a.out_row0) * (uint32_t)pl.pw + 2u * (uint32_t)pair;
                            khg_st16(a.r + o, rgb[0]);
                            khg_st16(a.g + o, rgb[1]);
                            khg_st16(a.b + o, rgb[2]);
                        }
                    }
                }
            }
        }
    }
}

// Launch kinds: level 2 of the three components into their LL buffers; level 1 of the three components into f64
// planes (the plane outputs of DecompressImage); level 1 of the three components and the colour conversion into the
// RGB image or into a BMP pixel array.  (2 and 3 were the luma-only launches the fused ones replaced.)
enum { KHG_INV_PLANES2 = 0, KHG_INV_PLANES1 = 1, KHG_INV_RGB_FUSED = 4, KHG_INV_BGR_FUSED = 5 };
union KhgInvSmemAny {
    KhgInvSmem<1> l1;
    KhgInvSmem<2> l2;
    KhgInvFusedSmem f;
};

template <int KIND>
KHG_DEV void khg_inv_launch_body(const KhgInvArgs& a, int wg, KhgInvSmemAny& sh)
{
    int p = 0;
    while (p + 1 < a.nplanes && wg >= a.pl[p + 1].wg0) ++p;
    if constexpr (KIND == KHG_INV_PLANES2) khg_inv_wg_body<2>(a, a.pl[p], wg - a.pl[p].wg0, sh.l2);
    else if constexpr (KIND == KHG_INV_PLANES1) khg_inv_wg_body<1>(a, a.pl[p], wg - a.pl[p].wg0, sh.l1);
    else if constexpr (KIND == KHG_INV_RGB_FUSED) khg_inv_fused_wg_body<KHG_OUT_RGB>(a, wg, sh.f);
    else khg_inv_fused_wg_body<KHG_OUT_BGR>(a, wg, sh.f);
}
