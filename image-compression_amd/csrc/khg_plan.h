// khg_plan.h -- host-side launch planning: workspace layout, quantiser constants, kernel
// argument blocks and grid sizes.  Shared by the HIP host code and the test-only simulator
// so both drive the kernel bodies identically.
#pragma once
#include <math.h>
#include <string.h>

#include "khg_common.h"
#include "khg_forward.h"
#include "khg_huffman.h"

// Quantiser constants of one plane type (Compress.h:143-209 for luma, 223-226 for chroma).
// rows/cols are the LUMA dimensions.
static inline void khg_make_quant(int rows, int cols, int lvl, int chroma, KhgQuant* q)
{
    int h = rows, w = cols, h_half = h / 2, w_half = w / 2, h_qut = h / 4, w_qut = w / 4;
    int map[7][4] = {
        {0, h_qut, 0, w_qut},           {0, h_qut, w_qut, w_half}, {h_qut, h_half, 0, w_qut},
        {h_qut, h_half, w_qut, w_half}, {0, h_half, w_half, w},    {h_half, h, 0, w_half},
        {h_half, h, w_half, w}};
    for (int index = 0; index < 7; ++index) {
        double r = lvl == 0 ? 8 : lvl == 1 ? 10 : 12;
        double i = 2;
        double c = 8;
        double f = lvl == 0 ? 23 : lvl == 1 ? 230 : 2300;
        double tau = pow(2, r - c + i) * (1 + f / pow(2, 11));
        double step;
        if (index == 0) step = tau / pow(2, i);
        else if (index == 1 || index == 2) step = tau / pow(2, 2 - 1);
        else if (index == 3) step = tau / pow(2, 2 - 2);
        else if (index == 4 || index == 5) step = tau / pow(2, 1 - 1);
        else step = tau / pow(2, 1 - 2);
        int y0 = map[index][0], y1 = map[index][1], x0 = map[index][2], x1 = map[index][3];
        if (chroma) { y0 /= 2; y1 /= 2; x0 /= 2; x1 /= 2; }
        int alpha = y0 + x0;
        int beta = y1 + x1;
        double a = 1.5 / (double)(beta - alpha);
        double b = 0.5 - (double)(a * alpha);
        q->step[index] = step;
        q->a[index] = a;
        q->b[index] = b;
    }
}

// Image sizes the library takes: multiples of 8 (two DWT levels on 4:2:0 planes) with fewer than 2^31 pixels (raster
// indices, plane offsets and the header's bit lengths are 32-bit).
static inline int khg_dims_ok(int rows, int cols)
{
    return rows > 0 && cols > 0 && rows % 8 == 0 && cols % 8 == 0 && (long long)rows * cols < (1LL << 31);
}

// Device workspace of one context (all pointers into device memory, or host memory in the simulator).
struct KhgWorkspace {
    double* ll[3];        // level-1 LL quadrants: Y (rows/2 x cols/2), Cb, Cr (rows/4 x cols/4)
    int16_t* sym[3];      // quantised planes
    uint32_t* hist_cnt;   // [3][KHG_NBINS]
    uint32_t* hist_first; // [3][KHG_NBINS]
    int32_t* tab_cnt;     // [3][KHG_NBINS]
    int32_t* tab_sym;     // [3][KHG_NBINS]
    int32_t* tab_k;       // [3]
    uint64_t* codetab;    // [3][KHG_NBINS]
    uint32_t* big_scratch;// [3][10*KHG_NBINS]
    int32_t* tree_left;   // [3][2*KHG_NBINS]
    int32_t* tree_right;  // [3][2*KHG_NBINS]
    int32_t* tree_root;   // [3]
    uint32_t* chunk_bits; // [max chunks]
    uint64_t* chunk_off;  // [max chunks]
    uint64_t* chain;      // [KHG_CHAIN_HEAD + 2 * (max chunks + 3)]: state of the single-pass packer (khg_packscan_body)
    KhgResult* res;
};

static inline uint32_t khg_nchunks(uint32_t n) { return (n + KHG_CHUNK - 1) / KHG_CHUNK; }

// Chunks of one plane of prow x pcol symbols: KHG_CHUNK-symbol pieces of the raster, or, for a row-tiled image,
// of the runs of khg_chunk_info.
static inline uint32_t khg_plane_chunks(int prow, int pcol, int nranks)
{
    if (nranks <= 1) return khg_nchunks((uint32_t)((size_t)prow * pcol));
    const uint32_t hh = (uint32_t)prow / 2;
    return hh * 2 * khg_nchunks((uint32_t)pcol / 2) + hh * khg_nchunks((uint32_t)pcol);
}

// Byte sizes of the workspace arrays for images up to rows x cols.
struct KhgWorkspaceSizes {
    size_t ll[3], sym[3], hist, tab, codetab, big_scratch, tree, chunk_bits, chunk_off, chain;
};
static inline void khg_workspace_sizes(int rows, int cols, KhgWorkspaceSizes* s)
{
    size_t ny = (size_t)rows * cols, nc = ny / 4;
    s->ll[0] = nc * sizeof(double) + KHG_DUMP_BYTES;
    s->ll[1] = s->ll[2] = (nc / 4) * sizeof(double) + KHG_DUMP_BYTES;
    s->sym[0] = ny * sizeof(int16_t) + KHG_DUMP_BYTES;
    s->sym[1] = s->sym[2] = nc * sizeof(int16_t) + KHG_DUMP_BYTES;
    s->hist = 3 * (size_t)KHG_NBINS * sizeof(uint32_t);
    s->tab = 3 * (size_t)KHG_NBINS * sizeof(int32_t);
    s->codetab = 3 * (size_t)KHG_NBINS * sizeof(uint64_t);
    s->big_scratch = 3 * (size_t)10 * KHG_NBINS * sizeof(uint32_t);
    s->tree = 3 * (size_t)2 * KHG_NBINS * sizeof(int32_t);
    // plain chunking, or the row-run chunking of tiled images, whichever has more chunks
    size_t nch = khg_nchunks((uint32_t)ny) + 2 * khg_nchunks((uint32_t)nc);
    size_t nch_tiled = khg_plane_chunks(rows, cols, 2) + 2 * (size_t)khg_plane_chunks(rows / 2, cols / 2, 2);
    if (nch_tiled > nch) nch = nch_tiled;
    s->chunk_bits = nch * sizeof(uint32_t);
    s->chunk_off = nch * sizeof(uint64_t);
    s->chain = (KHG_CHAIN_HEAD + 4 * (nch + 3)) * sizeof(uint64_t);  // entry and group words: generous
}

// One tile of a row-tiled image: rank t of n owns image rows [rows*t/n, rows*(t+1)/n), rows/n a multiple of 8
// so luma and chroma bands line up at both levels.  {0, 1} is the whole image.
struct KhgTile {
    int rank, nranks;
};
enum { KHG_TILE_HALO = 24 };  // image rows a tile needs beyond its own on each side (see khg_fwd_args)

static inline int khg_tile_ok(int rows, KhgTile t)
{
    return t.nranks >= 1 && t.nranks <= 8 && t.rank >= 0 && t.rank < t.nranks && rows % (8 * t.nranks) == 0;
}
static inline int khg_tile_row0(int rows, KhgTile t) { return rows / t.nranks * t.rank; }
static inline int khg_tile_row1(int rows, KhgTile t) { return rows / t.nranks * (t.rank + 1); }
// first image row a tile's input buffer must hold, and one past the last
static inline int khg_tile_in_row0(int rows, KhgTile t)
{
    int r = khg_tile_row0(rows, t) - KHG_TILE_HALO;
    return r < 0 ? 0 : r;
}
static inline int khg_tile_in_row1(int rows, KhgTile t)
{
    int r = khg_tile_row1(rows, t) + KHG_TILE_HALO;
    return r > rows ? rows : r;
}

// Plane of colour component comp (0 Y, 1 Cb, 2 Cr) at level 1 or 2 of a forward launch, without its grid.
// coef (optional) is the component's full f64 coefficient plane.
// Tiles: level 2 of a tile's band [R0/2, R1/2) of the LL quadrant reads LL rows R0/2-4 .. R1/2+4, so level 1 is
// run over plane rows R0-8 .. R1+8 (symbols emitted for R0 .. R1 only), which reads plane rows R0-12 .. R1+12:
// 12 luma rows, or 12 chroma rows = 24 image rows, beyond the tile on each side.  r, g, b point at image row
// khg_tile_in_row0().  Every output is a fixed expression of the inputs, so the recomputed rows are bit-exact.
static inline void khg_fwd_plane(KhgFwdPlane* pl, const KhgWorkspace& ws, int rows, int cols, int lvl, int comp, int level,
                                 double* coef, KhgTile tile, const double* src_plane)
{
    memset(pl, 0, sizeof *pl);
    const int prow = comp ? rows / 2 : rows, pcol = comp ? cols / 2 : cols;  // component plane
    if (level == 1) {
        pl->ph = prow; pl->pw = pcol;
        pl->src = src_plane;  // level 1 from a caller-supplied f64 plane instead of RGB (KHG_SRC_PLANE)
        pl->src_pitch = pcol;
    } else {
        pl->ph = prow / 2; pl->pw = pcol / 2;
        pl->src = ws.ll[comp];
        pl->src_pitch = pcol / 2;
    }
    pl->ll = ws.ll[comp];
    pl->ll_pitch = pcol / 2;
    pl->sym = ws.sym[comp];
    pl->sym_pitch = pcol;
    pl->sym_dump = (uint32_t)((size_t)prow * pcol * sizeof(int16_t));
    pl->ll_dump = (uint32_t)((size_t)(prow / 2) * (pcol / 2) * sizeof(double));
    pl->coef = coef;
    pl->coef_pitch = pcol;
    pl->hist_cnt = ws.hist_cnt + (size_t)comp * KHG_NBINS;
    pl->hist_first = ws.hist_first + (size_t)comp * KHG_NBINS;
    khg_make_quant(rows, cols, lvl, comp != 0, &pl->q);
    const int R0 = khg_tile_row0(prow, tile), R1 = khg_tile_row1(prow, tile);  // the tile's band of this component
    if (level == 1) {
        pl->row_begin = R0 - 8 < 0 ? 0 : R0 - 8;
        pl->row_end = R1 + 8 > prow ? prow : R1 + 8;
        pl->emit_k0 = R0 / 2;
        pl->emit_k1 = R1 / 2;
    } else {
        pl->row_begin = R0 / 2;
        pl->row_end = R1 / 2;
        pl->emit_k0 = pl->row_begin / 2;
        pl->emit_k1 = pl->row_end / 2;
    }
}

// Workgroups of the chip that a transform launch is planned for: 4 per CU (4 waves per SIMD) on 256 CUs, for the
// kernels with one and with two sample pairs per lane alike (both are compiled for at most 128 registers).
#ifndef KHG_FWD_SLOTS
#define KHG_FWD_SLOTS 1024
#endif
#ifndef KHG_FWD_SLOTS_L2
#define KHG_FWD_SLOTS_L2 512
#endif
#ifndef KHG_FWD_SLOTS_LP2
#define KHG_FWD_SLOTS_LP2 1024
#endif

// One group of planes of a transform launch that share a grid (Y alone, Cb with Cr, ...): pw columns, `span` rows
// to transform, `weight` = cost of one of its workgroups per row, relative to the other groups of the launch.  The
// planner fills in the rest.
struct KhgGridGroup {
    int pw, span, weight;
    int strip_pairs;  // output pairs of one wave's strip
    int seg_rows, nseg, nstrip_wg, wg0;
};

enum { KHG_SEG_WARMUP = 8 };  // rows a workgroup transforms before its segment's first output row

// Grid of a transform launch.  A workgroup walks a segment of its rows plus KHG_SEG_WARMUP rows, so segments should
// be as tall as the chip allows; every workgroup of the launch is resident at once (`slots`) and the launch lasts as
// long as its slowest one, so the segment heights are the ones at which every group's workgroups cost the same,
// (rows + warm-up) x weight, and all of them together just fit the chip, within 16 .. KHG_MAX_SEG_ROWS rows.
// Returns the number of workgroups.
static inline int khg_plan_groups(KhgGridGroup* g, int ngroup, int slots)
{
    for (int i = 0; i < ngroup; ++i) {
        const int nstrips = (g[i].pw / 2 + g[i].strip_pairs - 1) / g[i].strip_pairs;
        g[i].nstrip_wg = (nstrips + KHG_WG_STRIPS - 1) / KHG_WG_STRIPS;
    }
    for (long T = 1;; ++T) {  // cost per workgroup
        long wgs = 0;
        bool all_max = true;
        for (int i = 0; i < ngroup; ++i) {
            const int span = g[i].span;
            int seg = (int)((T / g[i].weight - KHG_SEG_WARMUP) & ~1L);
            if (seg < 16) seg = 16;
            if (seg > KHG_MAX_SEG_ROWS) seg = KHG_MAX_SEG_ROWS;
            if (seg > span) seg = span;
            if (seg < KHG_MAX_SEG_ROWS && seg < span) all_max = false;  // this group could still take taller segments
            const int nseg = (span + seg - 1) / seg;
            g[i].nseg = nseg;
            g[i].seg_rows = 2 * ((span / 2 + nseg - 1) / nseg);  // the tallest of nseg even shares
            wgs += (long)g[i].nstrip_wg * nseg;
        }
        if (wgs <= slots || all_max) break;
    }
    int wg = 0;
    for (int i = 0; i < ngroup; ++i) {
        g[i].wg0 = wg;
        wg += g[i].nstrip_wg * g[i].nseg;
    }
    return wg;
}

// Grid of a forward launch: group i starts at plane first[i] and holds count[i] planes.
static inline void khg_fwd_plan_grid(KhgFwdArgs* a, int ngroup, const int* first, const int* count, const int* weight,
                                     int slots = 0)
{
    KhgGridGroup g[3];
    for (int i = 0; i < ngroup; ++i) {
        const KhgFwdPlane& pl = a->pl[first[i]];
        g[i].pw = pl.pw; g[i].span = pl.row_end - pl.row_begin; g[i].weight = weight[i];
        g[i].strip_pairs = KHG_STRIP_PAIRS_LP(count[i] == 2 ? a->lpc : a->lp);
    }
    a->tap = 0;
    for (int p = 0; p < a->nplanes; ++p) a->tap |= a->pl[p].coef != nullptr;
    a->nwg = khg_plan_groups(g, ngroup, slots ? slots : a->lp == 2 ? KHG_FWD_SLOTS_LP2 : KHG_FWD_SLOTS);
    for (int i = 0; i < ngroup; ++i)
        for (int j = 0; j < count[i]; ++j) {
            KhgFwdPlane& o = a->pl[first[i] + j];
            o.seg_rows = g[i].seg_rows; o.nseg = g[i].nseg; o.nstrip_wg = g[i].nstrip_wg; o.wg0 = g[i].wg0;
        }
}

static inline void khg_fwd_args_common(KhgFwdArgs* a, const KhgWorkspace& ws, const uint8_t* r, const uint8_t* g,
                                       const uint8_t* b, int rows, int cols, int lvl, KhgTile tile)
{
    memset(a, 0, sizeof *a);
    a->r = r; a->g = g; a->b = b;
    a->img_w = cols;
    a->in_row0 = khg_tile_in_row0(rows, tile);
    a->in_rows = khg_tile_in_row1(rows, tile) - a->in_row0;
    a->status = &ws.res->status;
    a->lvl = lvl;
}

// Level 1 of all three components from u8 RGB in one launch (KHG_FWD_RGB): luma workgroups, then workgroups that
// carry Cb and Cr together.  luma_only: just the luma workgroups (a measurement aid).
static inline KhgFwdArgs khg_fwd_args_rgb(const KhgWorkspace& ws, const uint8_t* r, const uint8_t* g, const uint8_t* b,
                                          int rows, int cols, int lvl, double* const coef[3],
                                          KhgTile tile = KhgTile{0, 1}, bool luma_only = false, int bgr_stride = 0)
{
    KhgFwdArgs a;
    khg_fwd_args_common(&a, ws, r, g, b, rows, cols, lvl, tile);
    a.bgr_stride = bgr_stride;  // nonzero: r is the pixel array of a 24-bit BMP (KHG_FWD_BGR launch)
    for (int p = 0; p < 3; ++p) khg_fwd_plane(&a.pl[p], ws, rows, cols, lvl, p, 1, coef ? coef[p] : nullptr, tile, nullptr);
    // a row of a chroma workgroup (Cb and Cr together, half the strips' pixels each) costs 2.25 luma rows (measured)
    const int first[2] = {0, 1}, count[2] = {1, 2}, weight[2] = {4, 9};
    a.nplanes = luma_only ? 1 : 3;
    a.lp = khg_fwd_lp_for(a);
    a.lpc = a.lp == 2 ? KHG_FWD_LPC : 1;
    a.lhist = 256;
    khg_fwd_plan_grid(&a, luma_only ? 1 : 2, first, count, weight);
    return a;
}

// Level 1 from caller-supplied f64 Y / Cb / Cr planes (KHG_FWD_PLANES1, ycbcr != null) or level 2 from the LL
// quadrants that level 1 left in the workspace (KHG_FWD_PLANES2): the three components in one launch.
static inline KhgFwdArgs khg_fwd_args_planes(const KhgWorkspace& ws, int rows, int cols, int lvl, int level,
                                             double* const coef[3], KhgTile tile = KhgTile{0, 1},
                                             const double* const ycbcr[3] = nullptr)
{
    KhgFwdArgs a;
    khg_fwd_args_common(&a, ws, nullptr, nullptr, nullptr, rows, cols, lvl, tile);
    for (int p = 0; p < 3; ++p)
        khg_fwd_plane(&a.pl[p], ws, rows, cols, lvl, p, level, coef ? coef[p] : nullptr, tile, ycbcr ? ycbcr[p] : nullptr);
    const int first[3] = {0, 1, 2}, count[3] = {1, 1, 1}, weight[3] = {1, 1, 1};
    a.nplanes = 3;
    a.lp = a.lpc = khg_fwd_lp_for(a);
    a.lhist = level == 1 ? 256 : (lvl == 0 ? 2048 : 1024);  // Normal preset: luma LL symbols reach 1009 (255 * 4 / 1.011)
    // Level 2 is a quarter of the samples: spread over every slot of the chip a wave would walk six row pairs and
    // warm up its column pipelines over four more.
    khg_fwd_plan_grid(&a, 3, first, count, weight, level == 2 ? KHG_FWD_SLOTS_L2 : 0);
    return a;
}

static inline KhgTableArgs khg_table_args(const KhgWorkspace& ws)
{
    KhgTableArgs t;
    t.hist_cnt = ws.hist_cnt; t.hist_first = ws.hist_first;
    t.tab_cnt = ws.tab_cnt; t.tab_sym = ws.tab_sym; t.tab_k = ws.tab_k;
    t.codetab = ws.codetab; t.big_scratch = ws.big_scratch;
    t.tree_left = ws.tree_left; t.tree_right = ws.tree_right; t.tree_root = ws.tree_root;
    t.status = &ws.res->status;
    return t;
}

static inline KhgEncArgs khg_enc_args(const KhgWorkspace& ws, int rows, int cols, int lvl, uint8_t* out, size_t cap,
                                      KhgTile tile = KhgTile{0, 1})
{
    KhgEncArgs e;
    memset(&e, 0, sizeof e);
    uint32_t chunk = 0;
    for (int p = 0; p < 3; ++p) {
        e.prow[p] = p ? rows / 2 : rows;
        e.pcol[p] = p ? cols / 2 : cols;
        e.n[p] = (uint32_t)((size_t)e.prow[p] * e.pcol[p]);
        e.chunk0[p] = chunk;
        chunk += khg_plane_chunks(e.prow[p], e.pcol[p], tile.nranks);
    }
    e.nchunks = chunk;
    e.chain_entries = 3;
    for (int p = 0; p < 3; ++p) e.chain_entries += khg_chain_plane_entries((p < 2 ? e.chunk0[p + 1] : chunk) - e.chunk0[p]);
    e.rank = tile.rank;
    e.nranks = tile.nranks;
    e.write_meta = tile.rank == 0;
    for (int p = 0; p < 3; ++p) e.sym[p] = ws.sym[p];
    e.codetab = ws.codetab;
    e.chunk_bits = ws.chunk_bits; e.chunk_off = ws.chunk_off;
    e.chain = ws.chain;
    e.tab_cnt = ws.tab_cnt; e.tab_sym = ws.tab_sym; e.tab_k = ws.tab_k;
    e.out = out;
    e.cap = cap & ~(size_t)3;
    e.rows = rows; e.cols = cols; e.lvl = lvl;
    e.res = ws.res;
    return e;
}

// Upper bound of the .comp size: 36-byte header, streams, three full tables; multiple of 4.
// A Huffman code's mean length is below entropy + 1 <= log2(K) + 1 <= 15 bits for K <= 16384.
static inline size_t khg_comp_bound_impl(int rows, int cols)
{
    size_t n = (size_t)rows * cols * 3 / 2;
    size_t b = 36 + (n * 15 + 7) / 8 + 3 + 3 * (size_t)KHG_NBINS * 8;
    return (b + 3) & ~(size_t)3;
}
