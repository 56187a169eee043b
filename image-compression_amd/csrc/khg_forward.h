// khg_forward.h -- colour transform + one 2-D CDF 9/7 lifting level + quantiser + histogram.
//
// Lane-streaming layout.  A wave owns a strip of 64 * LP sample pairs of the plane being transformed and walks down a
// segment of its rows, two rows per trip; lane l holds LP neighbouring pairs (x[2p], x[2p+1]), p = first pair + LP l + a:
//   rows   : the four lifting steps of a row need the neighbouring pair's value once each (e[p+1], d1[p-1],
//            s1[p+1], d2[p-1]); inside a lane that is another register, at the lane's ends it comes from the
//            neighbouring LANE by a DPP wave shift, so a row is transformed where it was loaded, with no window
//            recomputed per thread, no staging tile and no barrier.  The two pairs at each end of a strip are halo
//            (their results are wrong and not used): 60 of 64 pairs are output with one pair per lane, 124 of 128
//            with two (and half the lane exchanges per sample).
//            Halo pairs that fall outside the plane are LOADED mirrored (whole-sample symmetric extension about the
//            first and the last sample: x[-i] = x[i], x[n-1+i] = x[n-1-i]).  Lifting preserves that symmetry and
//            a + b == b + a bit for bit, so the interior formula then yields exactly what the reference's edge
//            lines compute (DWT.h:18,26,34,42: x[n-1] += 2*a*x[n-2], x[0] += 2*a*x[1], and 2*a*x == a*(x+x)), and
//            the row pass has no edge case at all.  A mirrored pair is the memory pair (x[i], x[i+1]) read backwards,
//            so a lane carries its load column and where in the loaded window each of its samples sits.
//   columns: each pair's two packed output columns (low half p, high half w/2 + p) run the streaming lifting
//            pipeline of khg_fwd_step over the rows as they arrive, state in registers.  Finished coefficients are
//            emitted at once: level-1 LL scaled to an f64 buffer (input of level 2), everything else quantised
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
    KHG_STRIP_PAIRS = 60,        // output pairs per wave strip at one pair per lane (64 lanes minus two halo pairs per side)
    KHG_WG_STRIPS = 4,           // waves (adjacent strips) per workgroup
    KHG_MAX_SEG_ROWS = 120,      // rows per workgroup segment, at most
    KHG_RD_PAD = 4,              // the column pipeline also looks up the row pairs it only flushes
    KHG_TALLY_TRIPS = 16,        // a lane's 8-bit symbol tallies go to the workgroup histogram every so many trips
    KHG_DUMP_BYTES = 8192        // room behind a symbol plane and an LL buffer for the stores that have no output
};
// output pairs per wave strip with LP pairs per lane, and reciprocals per band that a workgroup tabulates
#define KHG_STRIP_PAIRS_LP(LP) (64 * (LP) - 4)
#define KHG_RD_N_LP(LP) (KHG_WG_STRIPS * KHG_STRIP_PAIRS_LP(LP) + 4 + KHG_MAX_SEG_ROWS / 2 + 2 * KHG_RD_PAD)
static_assert(3 * 2 * KHG_TALLY_TRIPS <= 255, "a lane's symbol tallies are 8-bit fields");

// One plane of a launch.
struct KhgFwdPlane {
    const double* src;         // KHG_SRC_PLANE: f64 source plane
    int src_pitch;
    int ph, pw;                // rows, cols of this transform
    double* ll;                // level 1: receives the LL quadrant (ph/2 x pw/2)
    int ll_pitch;
    int16_t* sym;              // symbol plane of this colour component
    int sym_pitch;
    uint32_t sym_dump, ll_dump;  // byte offsets behind the symbol plane / the LL buffer of KHG_DUMP_BYTES that take the
                                 // stores of lanes and trips without output (see khg_fwd_wg_body)
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
    int lp;                    // sample pairs per lane (1 or 2) of the luma / plane workgroups: which instantiation
                               // of the kernel the grid is planned for
    int lpc;                   // the same for the workgroups that carry Cb and Cr (RGB launches)
    int tap;                   // some plane has a coefficient tap (KhgFwdPlane::coef): the instantiation that stores them
    // RGB launch: [0] = Y, [1] = Cb, [2] = Cr (Cb and Cr share one grid: a lane carries both).
    // Plane launches: up to three independent planes.
    KhgFwdPlane pl[3];
    int lhist;                 // level 2: bins of the workgroup histogram in use (a power of two up to the capacity)
};

// Two pairs per lane need every plane of the launch to hold an even number of pairs (a lane's two pairs are then
// inside or outside the plane together, and their symbols and LL values are stored as one aligned word).
// Pairs per lane of the workgroups that carry Cb and Cr together.  With two they hold twice the column state of a
// luma workgroup (194 registers: two workgroups per CU instead of four), and the launch then took as long as with one
// pair per lane everywhere; with one, luma's two pairs fit the 128 registers of four workgroups per CU.
#ifndef KHG_FWD_LPC
#define KHG_FWD_LPC 1
#endif
static inline int khg_fwd_lp_for(const KhgFwdArgs& a)
{
#ifdef KHG_FWD_FORCE_LP1  // developer builds: the one-pair kernels everywhere
    return 1;
#endif
    for (int p = 0; p < a.nplanes; ++p)
        if ((a.pl[p].pw / 2) % 2) return 1;
    return 2;
}

// LDS of one workgroup.  Level 1 emits detail bands only (small symbols, three bands): 256 bins.  Level 2 also emits
// the LL band, whose symbols spread over hundreds of values and at the Normal preset (LL step 1.011) reach 1009: room
// for 2048 bins (more would cost a resident workgroup per CU that the launch plan counts on), of which a launch uses
// KhgFwdArgs::lhist.  Symbols outside the bins go to the global histogram one atomic at a time, and neighbouring LL
// symbols collide there: with 1024 bins the Normal preset's level 2 took 0.75 ms instead of 0.05.
template <int LEVEL, int NP, int LP>
struct KhgFwdSmem {
    enum { LHIST = LEVEL == 1 ? 256 : 2048, LHIST_BIAS = LHIST / 2, NBAND = LEVEL == 1 ? 3 : 4, RD_N = KHG_RD_N_LP(LP) };
    uint32_t cnt[NP][LHIST];    // symbols [-LHIST/2, LHIST/2)
    uint32_t first[NP][LHIST];
    double rdiv[NBAND * RD_N];  // band slot s at diagonal u: [s * RD_N + KHG_RD_PAD + u] (see khg_fwd_rdiv_entry)
};
union KhgFwdSmemAny {
    KhgFwdSmem<1, 1, 1> y1;
    KhgFwdSmem<1, 2, 1> c1;
    KhgFwdSmem<2, 1, 1> l21;
    KhgFwdSmem<1, 1, 2> y2;
    KhgFwdSmem<1, 2, 2> c2;
    KhgFwdSmem<2, 1, 2> l22;
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
#define KHG_SCALE_A (1 / 1.149604398)
KHG_DEV double khg_fwd_scale_even(double x)
{
    const double a = KHG_SCALE_A, ra = 1 / a;
    return khg_div_const(x, a, ra);
}
KHG_DEV double khg_fwd_scale_odd(double x) { const double a = KHG_SCALE_A; return x * a; }

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
KHG_DEV uint32_t khg_hi32(double v) { return (uint32_t)__double2hiint(v); }
// the low halves of a and b in one word (a in bits 0-15)
KHG_DEV uint32_t khg_pack16(int a, int b) { return __builtin_amdgcn_perm((uint32_t)b, (uint32_t)a, 0x05040100u); }
#else
#define KHG_LANE_NEXT(dst, src) \
    for (int t_ = 0; t_ < 256; ++t_) (dst)[t_] = (t_ & 63) == 63 ? (src)[t_] : (src)[t_ + 1];
#define KHG_LANE_PREV(dst, src) \
    for (int t_ = 0; t_ < 256; ++t_) (dst)[t_] = (t_ & 63) == 0 ? (src)[t_] : (src)[t_ - 1];
#define KHG_UNIFORM(x) (x)
static inline int khg_f2i(double n) { return n >= 2147483647.0 ? 2147483647 : (n <= -2147483648.0 ? (-2147483647 - 1) : (int)n); }
static inline uint32_t khg_hi32(double v) { uint64_t u; memcpy(&u, &v, 8); return (uint32_t)(u >> 32); }
static inline uint32_t khg_pack16(int a, int b) { return ((uint32_t)a & 0xFFFFu) | ((uint32_t)b << 16); }
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
template <int SRC, int LP>
struct KhgFwdRaw {
    // Per pair -- Y: r, g, b of row a, then of row b, two pixels per 16-bit load.  C: r, g, b of the two pixel rows,
    // four pixels per 32-bit load (bytes 0 and 2 are the two chroma sites).  PLANE: the pair of row a, the pair of
    // row b.  Y_BGR: the six bytes of two pixels as three 16-bit loads per row.  C_BGR: two 32-bit loads per row, one
    // per site.  Y with two pairs per lane: one 32-bit load per plane and row holds both pairs.
    uint32_t w[SRC == KHG_SRC_PLANE ? 8 * LP : (SRC == KHG_SRC_Y ? 6 : 6 * LP)];
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

// What khg_fwd_row_offsets needs of the launch, read once before the loop (left in the argument block the values
// were fetched from it again on every trip, each fetch a wait).
struct KhgFwdRows {
    int ph, pitch, in_row0, in_rows;  // pitch: elements per row of the source (BGR: bytes)
};
template <int SRC>
KHG_DEV KhgFwdRows khg_fwd_rows(const KhgFwdArgs& a, const KhgFwdPlane& pl)
{
    KhgFwdRows r;
    r.ph = KHG_UNIFORM(pl.ph);
    r.pitch = KHG_UNIFORM(SRC == KHG_SRC_PLANE ? pl.src_pitch : KHG_SRC_IS_BGR(SRC) ? a.bgr_stride : a.img_w);
    r.in_row0 = KHG_UNIFORM(a.in_row0);
    r.in_rows = KHG_UNIFORM(a.in_rows);
    KHG_KEEP_SCALAR(r.ph);
    KHG_KEEP_SCALAR(r.pitch);
    KHG_KEEP_SCALAR(r.in_row0);
    KHG_KEEP_SCALAR(r.in_rows);
    return r;
}

// Offsets of the two source rows of row pair j (plane rows 2j and 2j+1, reflected into the plane).  Nearly every
// trip's rows are plain rows of the plane and of the r/g/b band: that case is a few scalar instructions.
template <int SRC>
KHG_DEV void khg_fwd_row_offsets(const KhgFwdArgs& a, const KhgFwdPlane& pl, const KhgFwdRows& rw, int j, uint32_t& oa,
                                 uint32_t& ob)
{
    const int ya = 2 * j, ph = rw.ph;
    if (ya >= 0 && ya + 1 < ph) {
        if (SRC == KHG_SRC_PLANE) {
            oa = (uint32_t)ya * (uint32_t)rw.pitch;
            ob = oa + (uint32_t)rw.pitch;
            return;
        }
        const int f = KHG_SRC_IS_LUMA(SRC) ? 1 : 2;
        const int ra = f * ya - rw.in_row0, rb = ra + f;
        if (ra >= 0 && rb < rw.in_rows) {
            if (KHG_SRC_IS_BGR(SRC)) {
                oa = (uint32_t)(rw.in_rows - 1 - ra) * (uint32_t)rw.pitch;
                ob = (uint32_t)(rw.in_rows - 1 - rb) * (uint32_t)rw.pitch;
            } else {
                oa = (uint32_t)ra * (uint32_t)rw.pitch;
                ob = (uint32_t)rb * (uint32_t)rw.pitch;
            }
            return;
        }
    }
    oa = KHG_UNIFORM(khg_fwd_row_offset<SRC>(a, pl, khg_reflect(ya, ph)));
    ob = KHG_UNIFORM(khg_fwd_row_offset<SRC>(a, pl, khg_reflect(ya + 1, ph)));
}

// Where a lane's pairs (x[2p], x[2p+1]) are read from: all samples reflected into the row (p may lie up to two pairs
// outside the plane, and anywhere for lanes beyond the last strip, whose loads are not used).  The two reflected
// columns of a pair are always neighbours; col[a] is the lower one and she / sho say where in the loaded window the
// even and the odd sample sit.
template <int LP>
struct KhgLaneSrc {
    uint32_t col[LP];  // column offset of the load in elements of the source (planar RGB: pixels; BGR: bytes)
    uint32_t she[LP];  // planar RGB: bit position of the even sample's byte in the loaded word; others: nonzero = swapped
    uint32_t sho[LP];  // planar RGB: bit position of the odd sample's byte
};

template <int SRC, int LP>
KHG_DEV KhgLaneSrc<LP> khg_fwd_lane_src(int p0, int pw)
{
    KhgLaneSrc<LP> l;
    if (SRC == KHG_SRC_Y && LP == 2) {
        // one 4-byte window for the lane's four samples: reflection moves by one column per sample, so they lie
        // within four columns; the window is pulled back into the row at its right end (pw >= 4)
        int r[4], lo = pw;
        for (int i = 0; i < 4; ++i) {
            r[i] = khg_reflect(2 * p0 + i, pw);
            lo = r[i] < lo ? r[i] : lo;
        }
        if (lo > pw - 4) lo = pw - 4;
        for (int a = 0; a < LP; ++a) {
            l.col[a] = (uint32_t)lo;
            l.she[a] = 8u * (uint32_t)(r[2 * a] - lo);
            l.sho[a] = 8u * (uint32_t)(r[2 * a + 1] - lo);
        }
        return l;
    }
    for (int a = 0; a < LP; ++a) {
        const int p = p0 + a;
        const int ec = khg_reflect(2 * p, pw), oc = khg_reflect(2 * p + 1, pw);
        const bool swap = ec > oc;
        const uint32_t lo = (uint32_t)(swap ? oc : ec);
        const uint32_t sh = SRC == KHG_SRC_Y ? 8u : 16u;  // distance of the pair's samples in the word: chroma sites are
                                                          // every second pixel (Colorspace.h:16-21)
        l.col[a] = SRC == KHG_SRC_C ? 2u * lo : SRC == KHG_SRC_Y_BGR ? 3u * lo : SRC == KHG_SRC_C_BGR ? 6u * lo : lo;
        l.she[a] = swap ? sh : 0u;
        l.sho[a] = swap ? 0u : sh;
    }
    return l;
}

// Two rows of the plane at row offsets oa, ob (khg_fwd_row_offsets), at the lane's columns.  Offsets are 32-bit: a
// plane has fewer than 2^31 samples.  Loads of 2 / 4 / 16 bytes at any 1 / 2 / 8-byte boundary.
template <int SRC, int LP>
KHG_DEV void khg_fwd_fetch(const KhgFwdArgs& a, const KhgFwdPlane& pl, KhgFwdRaw<SRC, LP>& raw, uint32_t oa0, uint32_t ob0,
                           const KhgLaneSrc<LP>& ls)
{
    if constexpr (SRC == KHG_SRC_Y && LP == 2) {
        const uint32_t oa = oa0 + ls.col[0], ob = ob0 + ls.col[0];
        raw.w[0] = khg_ld32(a.r + oa); raw.w[1] = khg_ld32(a.g + oa); raw.w[2] = khg_ld32(a.b + oa);
        raw.w[3] = khg_ld32(a.r + ob); raw.w[4] = khg_ld32(a.g + ob); raw.w[5] = khg_ld32(a.b + ob);
    } else {
        KHG_UNROLL
        for (int p = 0; p < LP; ++p) {
            const uint32_t oa = oa0 + ls.col[p], ob = ob0 + ls.col[p];
            uint32_t* w = &raw.w[(SRC == KHG_SRC_PLANE ? 8 : 6) * p];
            if constexpr (SRC == KHG_SRC_PLANE) {
                __builtin_memcpy(&w[0], pl.src + oa, 16);
                __builtin_memcpy(&w[4], pl.src + ob, 16);
            } else if constexpr (SRC == KHG_SRC_Y) {
                w[0] = khg_ld16(a.r + oa); w[1] = khg_ld16(a.g + oa); w[2] = khg_ld16(a.b + oa);
                w[3] = khg_ld16(a.r + ob); w[4] = khg_ld16(a.g + ob); w[5] = khg_ld16(a.b + ob);
            } else if constexpr (SRC == KHG_SRC_Y_BGR) {  // B G R B G R of two neighbouring pixels
                w[0] = khg_ld16(a.r + oa); w[1] = khg_ld16(a.r + oa + 2); w[2] = khg_ld16(a.r + oa + 4);
                w[3] = khg_ld16(a.r + ob); w[4] = khg_ld16(a.r + ob + 2); w[5] = khg_ld16(a.r + ob + 4);
            } else if constexpr (SRC == KHG_SRC_C_BGR) {  // B G R (and one more byte) of two pixels two columns apart
                w[0] = khg_ld32(a.r + oa); w[1] = khg_ld32(a.r + oa + 6);
                w[2] = khg_ld32(a.r + ob); w[3] = khg_ld32(a.r + ob + 6);
            } else {
                w[0] = khg_ld32(a.r + oa); w[1] = khg_ld32(a.g + oa); w[2] = khg_ld32(a.b + oa);
                w[3] = khg_ld32(a.r + ob); w[4] = khg_ld32(a.g + ob); w[5] = khg_ld32(a.b + ob);
            }
        }
    }
}

// Samples (even, odd) of pair p of the lane, row rr (0: a, 1: b), channel n (Y / plane: 0; chroma: 0 = Cb, 1 = Cr) as
// the f64 values the transform runs on.  Main.cpp:200-202 widens u8 to double; Colorspace.h:11,19,20: products and
// sums left to right.
template <int SRC, int LP>
KHG_DEV void khg_fwd_samples(const KhgFwdRaw<SRC, LP>& raw, const KhgLaneSrc<LP>& ls, int p, int rr, int n, double& e,
                             double& o)
{
    const uint32_t she = ls.she[p], sho = ls.sho[p];
    if constexpr (SRC == KHG_SRC_PLANE) {
        double lo, hi;
        __builtin_memcpy(&lo, &raw.w[8 * p + 4 * rr], 8);
        __builtin_memcpy(&hi, &raw.w[8 * p + 4 * rr + 2], 8);
        e = she ? hi : lo;
        o = she ? lo : hi;
    } else if constexpr (KHG_SRC_IS_BGR(SRC)) {
        const uint32_t* w = &raw.w[6 * p];
        uint32_t pa, pb;  // (B, G, R) of the lower and of the upper pixel in bits 0-23
        if (SRC == KHG_SRC_Y_BGR) {
            const uint32_t w0 = w[3 * rr], w1 = w[3 * rr + 1], w2 = w[3 * rr + 2];
            pa = w0 | (w1 << 16);
            pb = (w1 >> 8) | (w2 << 8);
        } else {
            pa = w[2 * rr];
            pb = w[2 * rr + 1];
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
        e = she ? vb : va;
        o = she ? va : vb;
    } else {
        const uint32_t* w = &raw.w[SRC == KHG_SRC_Y && LP == 2 ? 0 : 6 * p];
        const uint32_t r = w[3 * rr], g = w[3 * rr + 1], b = w[3 * rr + 2];
        const double R0 = (double)((r >> she) & 255u), G0 = (double)((g >> she) & 255u), B0 = (double)((b >> she) & 255u);
        const double R1 = (double)((r >> sho) & 255u), G1 = (double)((g >> sho) & 255u), B1 = (double)((b >> sho) & 255u);
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

// ---- quantiser (Compress.h:218-219) and symbol table counts (Huffman.h:110-139).
//
// Per-thread tally of the symbols -1, 0, +1, which make up almost all detail-band symbols: counting them in a
// register keeps 64 lanes from hammering one LDS word.  `cnt`: 8-bit counters, zeros in byte 0, -1 in byte 1, +1 in
// byte 3 (a symbol's unit is one v_alignbyte of its two low bits), moved to the workgroup histogram every
// KHG_TALLY_TRIPS trips.  `seen`: one bit per (symbol, output stream), bit 4 t + (q & 3).  A lane's raster indices
// rise within each of its two output streams (rows k: low-half column, then high-half column; rows hh+k: the same),
// so only the first hit of a symbol in a stream can lower the workgroup's first-appearance index: one LDS atomic per
// bit, ever.  And once the workgroup histogram holds all three symbols at positions before the row a wave works on,
// nothing that wave still emits can lower them: from then on (`done`) the wave only counts.

// Unit of symbol q in -1..1 in the 8-bit tallies (for other q: byte 2 or an alias of these, unused).
#ifndef KHG_SIM
KHG_DEV uint32_t khg_hot_unit(int q) { return __builtin_amdgcn_alignbyte(1u, 1u, (uint32_t)q); }
#else
KHG_DEV uint32_t khg_hot_unit(int q) { return (uint32_t)(0x0000000100000001ull >> (8u * ((uint32_t)q & 3u))); }
#endif
KHG_DEV uint32_t khg_hot_seen(int q, int stream) { return (1u << (4 * stream)) << ((uint32_t)q & 3u); }

// q = round(c / div) for c = x scaled by the column pass (DWT.h:45-50: even samples x / a, odd samples x * a), from
// rdiv = RN(1/div) with the scale folded in (khg_fwd_rdiv_entry), when that is safe: t = x * rdiv is within a few
// 2^-53 (relative) of the exact quotient, far less than 1e-5 for every |t| that can become a symbol, so both round to
// the same integer unless t sits within 1e-5 of a half-integer (then, and for NaN and infinity, the caller scales and
// divides as the reference does).  Away from ties round-to-nearest-even and round-half-away agree.  A huge finite t
// passes and saturates q, which is out of the symbol range and fails the job like the division would.
// nhi: the high word of the rounded quotient as a double (|n| <= 1 exactly when its low 31 bits are <= 0x3FF00000).
KHG_DEV bool khg_quant_try(double x, double rdiv, int* q, uint32_t* nhi)
{
    const double t = x * rdiv;
    const double n = __builtin_rint(t);
    *q = khg_f2i(n);
    *nhi = khg_hi32(n);
    return fabs(t - n) < 0.49999;
}

// The reference's own division, for the coefficients khg_quant_try declines.
KHG_DEV int khg_quant_divide(const KhgFwdArgs& a, const KhgFwdPlane& pl, double c, int band, int x, int y)
{
    double n = round(c / khg_divisor(pl.q, band, x, y, a.lvl));
    if (!(n >= -(double)KHG_SYM_BIAS && n < (double)KHG_SYM_BIAS)) n = (double)KHG_SYM_BIAS;  // out of range: flagged later
    return (int)n;
}

// Table entry of band `band` at position (x, y): 1 / divisor, times the column pass's scale of that band's samples
// (bands whose rows are low-pass rows -- 0, 1 and 4 -- divide by a, the others multiply by it).  Level-1 LL values
// are stored, not quantised, and keep the exact scaling.
// khg_quant_try only needs it to a relative 1e-9 or so (an entry that is off declines more quotients or fewer,
// never gives another symbol), so the reciprocal is the hardware's estimate (about 23 bits) and two Newton steps
// instead of a division: a workgroup tabulates some 1600 to 2300 entries before its first row.
#ifndef KHG_SIM
KHG_DEV double khg_recip_approx(double d)
{
    double r = __builtin_amdgcn_rcp(d);
    r = khg_fma(khg_fma(-d, r, 1.0), r, r);
    return khg_fma(khg_fma(-d, r, 1.0), r, r);
}
#else
KHG_DEV double khg_recip_approx(double d) { return 1.0 / d; }
#endif
KHG_DEV double khg_fwd_rdiv_entry(const KhgQuant& q, int band, int x, int y, int lvl)
{
    const double rd = khg_recip_approx(khg_divisor(q, band, x, y, lvl));
    const bool even = band == 0 || band == 1 || band == 4;
    return even ? rd * (1 / KHG_SCALE_A) : rd * KHG_SCALE_A;
}

// Count one symbol that the straight-line path could not finish (the lane's row pair holds a symbol other than -1,
// 0, +1, or the first appearance of one of those in its stream).  Returns the symbol to store (0 when out of range).
template <int LHIST>
KHG_DEV int khg_count_rare(const KhgFwdArgs& a, const KhgFwdPlane& pl, uint32_t* lcnt, uint32_t* lfirst, uint32_t& cnt,
                           uint32_t& seen, int q, int stream, uint32_t raster)
{
    const int lh = LHIST == 256 ? 256 : a.lhist;  // bins in use, centred in the LHIST there is room for
    if ((unsigned)(q + KHG_SYM_BIAS) >= 2u * KHG_SYM_BIAS) {
        khg_set_status(a.status, -4);
        q = 0;  // stored and counted as a zero; the job fails with KHG_ERR_RANGE
    }
    if ((uint32_t)(q + 1) <= 2u) {
        cnt += khg_hot_unit(q);
        const uint32_t bit = khg_hot_seen(q, stream);
        if (!(seen & bit)) {
            seen |= bit;
            khg_lds_min(&lfirst[q + LHIST / 2], raster);
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

// A lane's 8-bit tallies into the workgroup histogram.
template <int LHIST>
KHG_DEV void khg_tally_flush(uint32_t* lcnt, uint32_t& cnt)
{
    const uint32_t c0 = cnt & 0xFFu, cm = (cnt >> 8) & 0xFFu, cp = cnt >> 24;
    if (c0) khg_lds_add(&lcnt[LHIST / 2], c0);
    if (cm) khg_lds_add(&lcnt[LHIST / 2 - 1], cm);
    if (cp) khg_lds_add(&lcnt[LHIST / 2 + 1], cp);
    cnt = 0;
}

// Finished, unscaled outputs (s2, d2) of row pair k of the low-half column (L) and high-half column (H) of the
// lane's LP pairs (the first is pair `pair`): scale, store / quantise, count.  rd[s][p]: reciprocal divisors per band
// slot and pair (khg_fwd_rdiv_entry); so, llo: byte offsets of position (k, pair) in the symbol plane and in the LL
// buffer; done: nothing this wave emits from here on can lower the first-appearance index of -1, 0 or +1; emit: the
// row pair's symbols are stored and counted (KhgFwdPlane::emit_k0, emit_k1); hh, hw: half the plane's rows and
// columns; right, down: byte offsets in the symbol plane from (k, pair) to (k, hw + pair) and to (hh + k, pair).
// Nothing is stored here: llv receives the level-1 LL values, words the symbols of each band slot (two pairs per
// lane: both in one word), and the result says whether the symbols are to be stored (see the caller).
template <int LEVEL, int LP, int LHIST, bool TAP>
KHG_DEV bool khg_col_emit(const KhgFwdArgs& a, const KhgFwdPlane& pl, uint32_t* lcnt, uint32_t* lfirst, uint32_t& cnt,
                          uint32_t& seen, bool done, bool emit, int hh, int hw, uint32_t right, uint32_t down, int pair,
                          int k, const double* s2L, const double* d2L, const double* s2H, const double* d2H,
                          const double (*rd)[LP], uint32_t so, double* llv, uint32_t* words)
{
    if (TAP && pl.coef) {
        KHG_UNROLL
        for (int p = 0; p < LP; ++p) {
            const uint32_t c0 = (uint32_t)k * (uint32_t)pl.coef_pitch + (uint32_t)(pair + p);
            const uint32_t c1 = (uint32_t)(hh + k) * (uint32_t)pl.coef_pitch + (uint32_t)(pair + p);
            pl.coef[c0] = khg_fwd_scale_even(s2L[p]);
            pl.coef[c0 + (uint32_t)hw] = khg_fwd_scale_even(s2H[p]);
            pl.coef[c1] = khg_fwd_scale_odd(d2L[p]);
            pl.coef[c1 + (uint32_t)hw] = khg_fwd_scale_odd(d2H[p]);
        }
    }
    if (LEVEL == 1) {  // LL feeds level 2 (Compress.h:112-114)
        KHG_UNROLL
        for (int p = 0; p < LP; ++p) llv[p] = khg_fwd_scale_even(s2L[p]);
    }
    if (!emit) return false;  // a row pair a tile transforms for its neighbour's LL values only
    // Bands of this level (Compress.h:154-160): LL (level 2 only: band 0), then the three detail bands -- high-half
    // column of low-pass row k, low-half and high-half column of high-pass row hh + k.
    enum { NB = LEVEL == 1 ? 3 : 4, B0 = LEVEL == 1 ? 4 : 0, D0 = LEVEL == 1 ? 0 : 1, NS = NB * LP };
    const uint32_t boff[4] = {0, right, down, down + right};  // of slot s: boff[s + 4 - NB]
    int q[NB][LP];
    uint32_t nhi[NB][LP];
    bool clear = true;
    KHG_UNROLL
    for (int s = 0; s < NB; ++s) {
        const int b4 = s + 4 - NB;  // 0 LL, 1 high-half of row k, 2 low-half of row hh + k, 3 high-half of row hh + k
        KHG_UNROLL
        for (int p = 0; p < LP; ++p) {
            const double x = b4 == 0 ? s2L[p] : b4 == 1 ? s2H[p] : b4 == 2 ? d2L[p] : d2H[p];
            clear = khg_quant_try(x, rd[s][p], &q[s][p], &nhi[s][p]) && clear;
        }
    }
    if (!clear) {  // near a tie, not finite: the reference's own scaling, division and rounding decide
        KHG_UNROLL
        for (int s = 0; s < NB; ++s) {
            const int b4 = s + 4 - NB;
            KHG_UNROLL
            for (int p = 0; p < LP; ++p) {
                const double x = b4 == 0 ? s2L[p] : b4 == 1 ? s2H[p] : b4 == 2 ? d2L[p] : d2H[p];
                int qi;
                uint32_t ni;
                if (!khg_quant_try(x, rd[s][p], &qi, &ni)) {
                    const double c = b4 <= 1 ? khg_fwd_scale_even(x) : khg_fwd_scale_odd(x);
                    q[s][p] = khg_quant_divide(a, pl, c, B0 + s, ((b4 & 1) ? hw : 0) + pair + p, (b4 >= 2 ? hh : 0) + k);
                }
            }
        }
    }
    if (LEVEL == 2) {  // LL symbols spread over hundreds of values: straight to the histograms
        KHG_UNROLL
        for (int p = 0; p < LP; ++p) {
            int v = q[0][p];
            if ((unsigned)(v + KHG_SYM_BIAS) >= 2u * KHG_SYM_BIAS) {
                khg_set_status(a.status, -4);
                v = 0;  // stored and counted as a zero; the job fails with KHG_ERR_RANGE
            }
            const uint32_t raster = (so + 2u * (uint32_t)p) >> 1;
            if ((unsigned)(v + a.lhist / 2) < (unsigned)a.lhist) {
                khg_lds_add(&lcnt[v + LHIST / 2], 1u);
                khg_lds_min(&lfirst[v + LHIST / 2], raster);
            } else {
                khg_atomic_add(&pl.hist_cnt[v + KHG_SYM_BIAS], 1u);
                khg_atomic_min(&pl.hist_first[v + KHG_SYM_BIAS], raster);
            }
            q[0][p] = v;
        }
    }
    // detail symbols: all of them -1, 0 or +1 (and none of them new to its stream): count in the register
    uint32_t any = 0, units = 0, sn = 0;
    KHG_UNROLL
    for (int s = D0; s < NB; ++s)
        KHG_UNROLL
        for (int p = 0; p < LP; ++p) {
            any |= nhi[s][p];
            units += khg_hot_unit(q[s][p]);
        }
    bool fast = clear && (any & 0x7FFFFFFFu) <= 0x3FF00000u;
    if (!done) {
        KHG_UNROLL
        for (int s = D0; s < NB; ++s)
            KHG_UNROLL
            for (int p = 0; p < LP; ++p) sn |= khg_hot_seen(q[s][p], s + 4 - NB >= 2);
        fast = fast && (sn & ~seen) == 0;
    }
    if (fast) {
        cnt += units;
    } else {
        KHG_UNROLL
        for (int s = D0; s < NB; ++s)
            KHG_UNROLL
            for (int p = 0; p < LP; ++p)
                q[s][p] = khg_count_rare<LHIST>(a, pl, lcnt, lfirst, cnt, seen, q[s][p], s + 4 - NB >= 2,
                                                (so + boff[s + 4 - NB] + 2u * (uint32_t)p) >> 1);
    }
    KHG_UNROLL
    for (int s = 0; s < NB; ++s) words[s] = LP == 2 ? khg_pack16(q[s][0], q[s][LP - 1]) : (uint32_t)q[s][0];
    return true;
}

// Have all of -1, 0, +1 appeared in the workgroup's histograms of the NP planes before raster index `before`?
template <int NP, int LHIST>
KHG_DEV bool khg_hot_done(uint32_t (*lfirst)[LHIST], uint32_t before)
{
    bool all = true;
    for (int n = 0; n < NP; ++n)
        for (int v = -1; v <= 1; ++v) all = all && lfirst[n][LHIST / 2 + v] < before;
    return all;
}

#if defined(KHG_TIMING) && !defined(KHG_SIM)
// developer build: per-workgroup start / end stamps (100 MHz), hardware id and shader-clock cycles of the last forward launch
__device__ uint32_t khg_dbg_fwd[8 * 8192];
#define KHG_FWD_STAMP(wg, slot, v) \
    if (threadIdx.x == 0 && (wg) < 8192) khg_dbg_fwd[8 * (wg) + (slot)] = (v)
#define KHG_FWD_STAMP_HERE(slot) KHG_FWD_STAMP(khg_xcd_remap((int)blockIdx.x, a.nwg), slot, (uint32_t)wall_clock64())
#else
#define KHG_FWD_STAMP(wg, slot, v)
#define KHG_FWD_STAMP_HERE(slot)
#endif

// One workgroup: KHG_WG_STRIPS adjacent strips (one per wave) over one segment of rows of NP planes of equal
// geometry (NP = 2: Cb and Cr from the same pixels), LP pairs per lane.  wgl = workgroup index within the plane's grid.
// PF: how many trips ahead the source rows are requested.  One covers the memory latency where a trip is long (level
// 1 from u8 pixels, four workgroups per CU); level 2 has short trips and few of them per wave, and asks two ahead.
template <int SRC, int LEVEL, int NP, int LP, bool TAP, int PF = 1>
KHG_DEV void khg_fwd_wg_body(const KhgFwdArgs& a, const KhgFwdPlane* pls, int wgl, KhgFwdSmem<LEVEL, NP, LP>& sh)
{
    typedef KhgFwdSmem<LEVEL, NP, LP> SM;
    enum { SP = KHG_STRIP_PAIRS_LP(LP) };
    const KhgFwdPlane& g = pls[0];  // geometry (shared by the NP planes)
    const int swg = wgl % g.nstrip_wg, seg = wgl / g.nstrip_wg;
    const int np = g.pw / 2;
    int r0, r1;
    khg_segment_rows(g.row_begin, g.row_end, g.nseg, seg, &r0, &r1);
    const int k0 = r0 / 2, k1 = r1 / 2;
    const int P0 = SP * KHG_WG_STRIPS * swg;  // first output pair of the workgroup
    const KhgFwdRows rw = khg_fwd_rows<SRC>(a, g);
    // loop invariants of the planes, read once
    int ek0 = KHG_UNIFORM(g.emit_k0), ek1 = KHG_UNIFORM(g.emit_k1);
    uint32_t sym_step = 2u * (uint32_t)KHG_UNIFORM(g.sym_pitch), ll_step = 8u * (uint32_t)KHG_UNIFORM(g.ll_pitch);
    KHG_KEEP_SCALAR(ek0);
    KHG_KEEP_SCALAR(ek1);
    KHG_KEEP_SCALAR(sym_step);
    KHG_KEEP_SCALAR(ll_step);
    // byte offsets in the symbol plane from position (k, pair) to (k, w/2 + pair) and to (h/2 + k, pair)
    uint32_t right = 2u * (uint32_t)(g.pw / 2), down = 2u * (uint32_t)(g.ph / 2) * (uint32_t)g.sym_pitch;
    char *symb[NP], *llb[NP];  // the planes' buffers and the dump areas behind them
    uint32_t symd[NP], lld[NP];
    for (int n = 0; n < NP; ++n) {
        symb[n] = (char*)pls[n].sym;
        llb[n] = (char*)pls[n].ll;
        symd[n] = KHG_UNIFORM(pls[n].sym_dump);
        lld[n] = KHG_UNIFORM(pls[n].ll_dump);
        KHG_KEEP_SCALAR(symb[n]);
        KHG_KEEP_SCALAR(llb[n]);
        KHG_KEEP_SCALAR(symd[n]);
        KHG_KEEP_SCALAR(lld[n]);
    }
    int hh_ = KHG_UNIFORM(g.ph / 2), hw_ = KHG_UNIFORM(g.pw / 2);
    KHG_KEEP_SCALAR(right);
    KHG_KEEP_SCALAR(down);
    KHG_KEEP_SCALAR(hh_);
    KHG_KEEP_SCALAR(hw_);

    KhgLift colL[NP][LP][KHG_NSLOT(256)], colH[NP][LP][KHG_NSLOT(256)];
    uint32_t cnts[NP][KHG_NSLOT(256)], seens[NP][KHG_NSLOT(256)];
    int dones[KHG_NSLOT(256)];     // the wave only counts -1, 0, +1 from here on (see khg_hot_done)
    KhgFwdRaw<SRC, LP> raws[PF][KHG_NSLOT(256)];  // the source rows of this trip and of the PF - 1 trips after it
    KhgLaneSrc<LP> lsrc[KHG_NSLOT(256)];
    int pairs[KHG_NSLOT(256)];     // the lane's first pair, clamped to the strip's output pairs
    bool outl[KHG_NSLOT(256)];     // the lane outputs its pairs
    uint32_t symo[KHG_NSLOT(256)], llo[KHG_NSLOT(256)];  // byte offsets of (k, pair) in the symbol plane / LL buffer
    // values of the row pass that cross an exchange: [row a / b][plane][pair]
    double E[2][NP][LP][KHG_NSLOT(256)], O[2][NP][LP][KHG_NSLOT(256)], D1[2][NP][LP][KHG_NSLOT(256)],
        S1[2][NP][LP][KHG_NSLOT(256)], D2[2][NP][LP][KHG_NSLOT(256)], X[2][NP][KHG_NSLOT(256)];

    KHG_FOR_THREADS(256)
    {
        if (a.prof && tid == 0) khg_atomic_max64(&a.prof->prof_not_start, ~khg_clock());
        const int lane = tid & 63, s = swg * KHG_WG_STRIPS + KHG_UNIFORM(tid >> 6);
        // strip s outputs pairs SP s .. SP s + SP - 1 (those inside the plane); its lanes hold pairs SP s - 2 .. SP s + SP + 1
        const int v0 = SP * s, v1 = v0 + SP < np ? v0 + SP : np;
        const int p = v0 - 2 + LP * lane;
        outl[KHG_SLOT] = p >= v0 && p < v1;  // a wave beyond the last strip loads and computes without output
        pairs[KHG_SLOT] = p < v0 ? v0 : (p > v0 + SP - LP ? v0 + SP - LP : p);
        lsrc[KHG_SLOT] = khg_fwd_lane_src<SRC, LP>(p, g.pw);
        // the first trip finishes row pair k0 - 4 (nothing is stored before k0; the offsets just run along)
        symo[KHG_SLOT] = 2u * ((uint32_t)(k0 - 4) * (uint32_t)g.sym_pitch + (uint32_t)pairs[KHG_SLOT]);
        llo[KHG_SLOT] = 8u * ((uint32_t)(k0 - 4) * (uint32_t)g.ll_pitch + (uint32_t)pairs[KHG_SLOT]);
        const KhgLift z = {0, 0, 0, 0, 0};
        for (int n = 0; n < NP; ++n) {
            for (int q = 0; q < LP; ++q) {
                colL[n][q][KHG_SLOT] = z;
                colH[n][q][KHG_SLOT] = z;
            }
            cnts[n][KHG_SLOT] = 0;
            seens[n][KHG_SLOT] = 0;
        }
        dones[KHG_SLOT] = 0;
        // rows of the first trip
        uint32_t foa, fob;
        for (int i = 0; i < PF; ++i) {
            khg_fwd_row_offsets<SRC>(a, g, rw, k0 - 2 + i, foa, fob);
            khg_fwd_fetch<SRC, LP>(a, g, raws[i][KHG_SLOT], foa, fob, lsrc[KHG_SLOT]);
        }
        const int lh = LEVEL == 1 ? (int)SM::LHIST : a.lhist, l0 = SM::LHIST_BIAS - lh / 2;  // the bins in use
        for (int n = 0; n < NP; ++n)
            for (int i = l0 + tid; i < l0 + lh; i += 256) {
                sh.cnt[n][i] = 0;
                sh.first[n][i] = 0xFFFFFFFFu;
            }
        // The divisor depends on the band and on x + y only (Compress.h:215-218); pair P0 + i at row pair k sits
        // on diagonal u = i + (k - k0) of each band this workgroup emits.
        const int hh = g.ph / 2, hw = g.pw / 2;
        const int nu = KHG_WG_STRIPS * SP + (k1 - k0);
        KHG_UNROLL
        for (int s2 = 0; s2 < SM::NBAND; ++s2) {
            const int band = LEVEL == 1 ? 4 + s2 : s2;
            const int by = (LEVEL == 1 ? (band >= 5) : (band >= 2)) ? hh : 0;
            const int bx = (LEVEL == 1 ? (band != 5) : (band & 1)) ? hw : 0;
            for (int u = tid; u < nu; u += 256)
                sh.rdiv[s2 * SM::RD_N + KHG_RD_PAD + u] = khg_fwd_rdiv_entry(g.q, band, bx + P0 + u, by + k0, a.lvl);
        }
    }
    KHG_BARRIER();
    KHG_FWD_STAMP_HERE(6);

    // Row pairs k0-2 .. k1+1 are fed; the column pipeline hands back pair j-2 when pair j goes in.
    const int jend = k1 + 1;
#ifndef KHG_SIM
    // a wave whose strip lies beyond the plane has nothing to do until the workgroup's last barrier
    const bool idle = SP * (swg * KHG_WG_STRIPS + KHG_UNIFORM((int)(threadIdx.x >> 6))) >= np;
#else
    const bool idle = false;
#endif
    auto trip = [&](const int j) __attribute__((always_inline)) {
        // source rows of the trip PF ahead (the same for every thread)
        uint32_t noa, nob;
        khg_fwd_row_offsets<SRC>(a, g, rw, j + PF, noa, nob);
        KHG_FOR_THREADS(256)
        {
            KHG_UNROLL
            for (int rr = 0; rr < 2; ++rr)
                KHG_UNROLL
                for (int n = 0; n < NP; ++n)
                    KHG_UNROLL
                    for (int p = 0; p < LP; ++p)
                        khg_fwd_samples<SRC, LP>(raws[0][KHG_SLOT], lsrc[KHG_SLOT], p, rr, n, E[rr][n][p][KHG_SLOT],
                                                 O[rr][n][p][KHG_SLOT]);
            // a later trip's rows: in flight during this trip's arithmetic (unconditional, also after the last
            // trip, where the rows are reflected back into the plane and the loads are simply not used: a
            // condition here would make the loads land in other registers and cost a wait and copies per trip)
            KHG_UNROLL
            for (int i = 0; i + 1 < PF; ++i) raws[i][KHG_SLOT] = raws[i + 1][KHG_SLOT];
            khg_fwd_fetch<SRC, LP>(a, g, raws[PF - 1][KHG_SLOT], noa, nob, lsrc[KHG_SLOT]);
        }
        // predict 1: d1[p] = o[p] + P1 * (e[p] + e[p+1])
        for (int rr = 0; rr < 2; ++rr)
            for (int n = 0; n < NP; ++n) KHG_LANE_NEXT(X[rr][n], E[rr][n][0])
        KHG_FOR_THREADS(256)
        {
            KHG_UNROLL
            for (int rr = 0; rr < 2; ++rr)
                KHG_UNROLL
                for (int n = 0; n < NP; ++n)
                    KHG_UNROLL
                    for (int p = 0; p < LP; ++p) {
                        const double nx = p + 1 < LP ? E[rr][n][p + 1 < LP ? p + 1 : 0][KHG_SLOT] : X[rr][n][KHG_SLOT];
                        D1[rr][n][p][KHG_SLOT] = O[rr][n][p][KHG_SLOT] + KHG_P1 * (E[rr][n][p][KHG_SLOT] + nx);
                    }
        }
        // update 1: s1[p] = e[p] + U1 * (d1[p-1] + d1[p])
        for (int rr = 0; rr < 2; ++rr)
            for (int n = 0; n < NP; ++n) KHG_LANE_PREV(X[rr][n], D1[rr][n][LP - 1])
        KHG_FOR_THREADS(256)
        {
            KHG_UNROLL
            for (int rr = 0; rr < 2; ++rr)
                KHG_UNROLL
                for (int n = 0; n < NP; ++n)
                    KHG_UNROLL
                    for (int p = 0; p < LP; ++p) {
                        const double pv = p > 0 ? D1[rr][n][p > 0 ? p - 1 : 0][KHG_SLOT] : X[rr][n][KHG_SLOT];
                        S1[rr][n][p][KHG_SLOT] = E[rr][n][p][KHG_SLOT] + KHG_U1 * (pv + D1[rr][n][p][KHG_SLOT]);
                    }
        }
        // predict 2: d2[p] = d1[p] + P2 * (s1[p] + s1[p+1])
        for (int rr = 0; rr < 2; ++rr)
            for (int n = 0; n < NP; ++n) KHG_LANE_NEXT(X[rr][n], S1[rr][n][0])
        KHG_FOR_THREADS(256)
        {
            KHG_UNROLL
            for (int rr = 0; rr < 2; ++rr)
                KHG_UNROLL
                for (int n = 0; n < NP; ++n)
                    KHG_UNROLL
                    for (int p = 0; p < LP; ++p) {
                        const double nx = p + 1 < LP ? S1[rr][n][p + 1 < LP ? p + 1 : 0][KHG_SLOT] : X[rr][n][KHG_SLOT];
                        D2[rr][n][p][KHG_SLOT] = D1[rr][n][p][KHG_SLOT] + KHG_P2 * (S1[rr][n][p][KHG_SLOT] + nx);
                    }
        }
        // update 2: s2[p] = s1[p] + U2 * (d2[p-1] + d2[p]); scale; columns
        for (int rr = 0; rr < 2; ++rr)
            for (int n = 0; n < NP; ++n) KHG_LANE_PREV(X[rr][n], D2[rr][n][LP - 1])
        const int k = j - 2;
        KHG_FOR_THREADS(256)
        {
            const bool out = outl[KHG_SLOT] && k >= k0 && k < k1;
            const int pair = pairs[KHG_SLOT];
            const double* rd = &sh.rdiv[KHG_RD_PAD + (pair - P0) + (k - k0)];
            double rdv[SM::NBAND][LP];
            KHG_UNROLL
            for (int s = 0; s < SM::NBAND; ++s)
                KHG_UNROLL
                for (int p = 0; p < LP; ++p) rdv[s][p] = rd[s * SM::RD_N + p];
            KHG_UNROLL
            for (int n = 0; n < NP; ++n) {
                double s2L[LP], d2L[LP], s2H[LP], d2H[LP];
                KHG_UNROLL
                for (int p = 0; p < LP; ++p) {
                    // row a is the even sample of both columns, row b the odd one
                    double lo[2], hi[2];
                    KHG_UNROLL
                    for (int rr = 0; rr < 2; ++rr) {
                        const double pv = p > 0 ? D2[rr][n][p > 0 ? p - 1 : 0][KHG_SLOT] : X[rr][n][KHG_SLOT];
                        const double s2 = S1[rr][n][p][KHG_SLOT] + KHG_U2 * (pv + D2[rr][n][p][KHG_SLOT]);
                        lo[rr] = khg_fwd_scale_even(s2);
                        hi[rr] = khg_fwd_scale_odd(D2[rr][n][p][KHG_SLOT]);
                    }
                    khg_fwd_step(colL[n][p][KHG_SLOT], lo[0], lo[1], s2L[p], d2L[p]);
                    khg_fwd_step(colH[n][p][KHG_SLOT], hi[0], hi[1], s2H[p], d2H[p]);
                }
                double llv[LP];
                uint32_t words[SM::NBAND];
                KHG_UNROLL
                for (int p = 0; p < LP; ++p) llv[p] = 0;
                KHG_UNROLL
                for (int s = 0; s < SM::NBAND; ++s) words[s] = 0;
                bool syms = false;
                if (out)
                    syms = khg_col_emit<LEVEL, LP, SM::LHIST, TAP>(a, pls[n], sh.cnt[n], sh.first[n], cnts[n][KHG_SLOT],
                                                              seens[n][KHG_SLOT], dones[KHG_SLOT] != 0, k >= ek0 && k < ek1, hh_, hw_,
                                                              right, down, pair, k, s2L, d2L,
                                                              s2H, d2H, rdv, symo[KHG_SLOT], llv, words);
                // The trip's stores, outside every condition: the compiler then knows how many stores follow the
                // loads of the next trip's rows and waits for those loads alone (behind a branch they made it wait
                // for every store of the trip, once per trip).  Lanes and trips without output store to a dump area
                // behind the buffer (two pairs per lane: an even pair of an even-pitched buffer, one aligned word /
                // 16 bytes).
                if (LEVEL == 1) {
                    const uint32_t o = out ? llo[KHG_SLOT] : lld[n] + (uint32_t)(8 * LP * tid);
                    __builtin_memcpy(KHG_ASSUME_ALIGNED(llb[n] + o, LP == 2 ? 16 : 8), llv, sizeof llv);
                }
                KHG_UNROLL
                for (int s = 0; s < SM::NBAND; ++s) {
                    const int b4 = s + 4 - SM::NBAND;
                    const uint32_t boff = b4 == 0 ? 0u : b4 == 1 ? right : b4 == 2 ? down : down + right;
                    const uint32_t o = syms ? symo[KHG_SLOT] + boff : symd[n] + (uint32_t)(1024 * s + (LP == 2 ? 4 : 2) * tid);
                    char* at = symb[n] + o;
                    if (LP == 2) __builtin_memcpy(KHG_ASSUME_ALIGNED(at, 4), &words[s], 4);
                    else *(int16_t*)at = (int16_t)words[s];
                }
            }
            symo[KHG_SLOT] += sym_step;
            llo[KHG_SLOT] += ll_step;
        }
        if (k >= k0 && k < k1) {
            const int t = k - k0 + 1;  // row pairs this segment has emitted
            if ((t & (KHG_TALLY_TRIPS - 1)) == 0) {
                KHG_FOR_THREADS(256)
                {
                    for (int n = 0; n < NP; ++n) khg_tally_flush<SM::LHIST>(sh.cnt[n], cnts[n][KHG_SLOT]);
                }
            }
            if ((t & (t - 1)) == 0) {  // after 1, 2, 4, ... row pairs: may the wave stop looking for first appearances?
                KHG_FOR_THREADS(256)
                {
                    if (!dones[KHG_SLOT]) {
                        const int dn = khg_hot_done<NP, SM::LHIST>(sh.first, (uint32_t)(k + 1) * (uint32_t)g.sym_pitch);
                        dones[KHG_SLOT] = KHG_UNIFORM(dn);
                    }
                }
            }
        }
    };
    if (!idle)
        for (int j = k0 - 2; j <= jend; ++j) trip(j);

    KHG_FWD_STAMP_HERE(7);
    KHG_FOR_THREADS(256)
    {
        for (int n = 0; n < NP; ++n) khg_tally_flush<SM::LHIST>(sh.cnt[n], cnts[n][KHG_SLOT]);
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

#ifndef KHG_FWD_PF_L2
#define KHG_FWD_PF_L2 2
#endif
// Launch kinds.
enum { KHG_FWD_RGB = 0, KHG_FWD_PLANES1 = 1, KHG_FWD_PLANES2 = 2, KHG_FWD_BGR = 3 };


// Workgroup `wg` of a launch: which plane it belongs to, and its body.
template <int KIND, int LP, int LPC, bool TAP>
KHG_DEV void khg_fwd_launch_body(const KhgFwdArgs& a, int wg, KhgFwdSmemAny& sh)
{
    wg = khg_xcd_remap(wg, a.nwg);
    KHG_FWD_STAMP(wg, 0, (uint32_t)wall_clock64());
    KHG_FWD_STAMP(wg, 2, __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)));  // HW_ID
    KHG_FWD_STAMP(wg, 3, __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)));  // XCC_ID
    KHG_FWD_STAMP(wg, 4, (uint32_t)__builtin_amdgcn_s_memtime());
    if constexpr (KIND == KHG_FWD_RGB) {
        if (a.nplanes < 2 || wg < a.pl[1].wg0)
            khg_fwd_wg_body<KHG_SRC_Y, 1, 1, LP, TAP>(a, &a.pl[0], wg, *(KhgFwdSmem<1, 1, LP>*)&sh);
        else khg_fwd_wg_body<KHG_SRC_C, 1, 2, LPC, TAP>(a, &a.pl[1], wg - a.pl[1].wg0, *(KhgFwdSmem<1, 2, LPC>*)&sh);
    } else if constexpr (KIND == KHG_FWD_BGR) {
        if (a.nplanes < 2 || wg < a.pl[1].wg0)
            khg_fwd_wg_body<KHG_SRC_Y_BGR, 1, 1, LP, TAP>(a, &a.pl[0], wg, *(KhgFwdSmem<1, 1, LP>*)&sh);
        else khg_fwd_wg_body<KHG_SRC_C_BGR, 1, 2, LPC, TAP>(a, &a.pl[1], wg - a.pl[1].wg0, *(KhgFwdSmem<1, 2, LPC>*)&sh);
    } else {
        int p = 0;
        while (p + 1 < a.nplanes && wg >= a.pl[p + 1].wg0) ++p;
        if constexpr (KIND == KHG_FWD_PLANES1)
            khg_fwd_wg_body<KHG_SRC_PLANE, 1, 1, LP, TAP>(a, &a.pl[p], wg - a.pl[p].wg0, *(KhgFwdSmem<1, 1, LP>*)&sh);
        else khg_fwd_wg_body<KHG_SRC_PLANE, 2, 1, LP, TAP, KHG_FWD_PF_L2>(a, &a.pl[p], wg - a.pl[p].wg0, *(KhgFwdSmem<2, 1, LP>*)&sh);
    }
    KHG_FWD_STAMP(wg, 1, (uint32_t)wall_clock64());
    KHG_FWD_STAMP(wg, 5, (uint32_t)__builtin_amdgcn_s_memtime());
}
