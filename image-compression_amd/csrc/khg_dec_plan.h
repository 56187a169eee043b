// khg_dec_plan.h -- host-side launch planning of the decode path (shared by the HIP host
// code and the test-only simulator).
#pragma once
#include "khg_plan.h"
#include "khg_decode.h"
#include "khg_inverse.h"
#include "khg_metrics.h"

static inline KhgDecArgs khg_dec_args(const KhgWorkspace& ws, const KhgDecWorkspace& dws, const uint8_t* comp,
                                      size_t len, int rows, int cols)
{
    KhgDecArgs d;
    memset(&d, 0, sizeof d);
    d.comp = comp;
    d.len = len;
    d.rows = rows;
    d.cols = cols;
    d.hdr = dws.hdr;
    d.res = ws.res;
    d.tab_cnt = ws.tab_cnt; d.tab_sym = ws.tab_sym;
    d.tree_left = ws.tree_left; d.tree_right = ws.tree_right; d.tree_root = ws.tree_root;
    d.big_scratch = ws.big_scratch;
    d.lut = dws.lut;
    d.lut2 = dws.lut2;
    uint32_t ny = (uint32_t)((size_t)rows * cols), nc = ny / 4;
    d.n[0] = ny; d.n[1] = nc; d.n[2] = nc;
    uint32_t blk = 0;
    for (int p = 0; p < 3; ++p) {
        d.blk0[p] = blk;
        blk += khg_max_blocks(d.n[p]);
        d.sym[p] = ws.sym[p];
    }
    d.nblocks = blk;
    d.blk = dws.blk;
    d.walk = dws.walk;
    return d;
}

// ---- row-tiled decode with the entropy decode split over the ranks.
// Active workgroups of each plane's stream, from the first 36 bytes of the file (Reader.h:5-23).  Returns 0 when
// the header is unusable (the device-side header check then fails the job).
static inline int khg_tile_active_blocks(const uint8_t hdr36[36], uint32_t nb[3])
{
    for (int p = 0; p < 3; ++p) {
        const uint32_t len = (uint32_t)hdr36[4 * p] | ((uint32_t)hdr36[4 * p + 1] << 8) | ((uint32_t)hdr36[4 * p + 2] << 16) |
                             ((uint32_t)hdr36[4 * p + 3] << 24);
        const uint32_t pad = hdr36[32 + p];
        if (len > 0x7FFFFFFFu || pad > 7 || pad > len) { nb[0] = nb[1] = nb[2] = 0; return 0; }
        nb[p] = khg_dec_active_blocks(len - pad);
    }
    return 1;
}

// Raster ranges [first, last) of plane `comp`'s symbols that the inverse transform of tile t reads: whole rows,
// up to four ranges (level 1: packed rows k and hh + k around the tile's band; level 2: the same inside the top
// left quadrant).  A superset of what the kernels touch (the left halves of the level-1 rows ride along).
static inline int khg_tile_need_ranges_impl(int rows, int cols, KhgTile t, int comp, uint32_t out[4][2])
{
    const int prow = comp ? rows / 2 : rows, pcol = comp ? cols / 2 : cols, hh = prow / 2, hh2 = hh / 2;
    const int R0 = khg_tile_row0(prow, t), R1 = khg_tile_row1(prow, t);
    // level 1 reconstructs rows [R0, R1): trips k0 - 2 .. k1 + 2 read packed rows k (reflected into the plane)
    int a1 = R0 / 2 - 4, b1 = R1 / 2 + 4;
    if (a1 < 0) a1 = 0;
    if (b1 > hh) b1 = hh;
    // level 2 reconstructs rows [rb, re) of the quadrant (khg_inv_plane)
    int rb = R0 / 2 - 4 < 0 ? 0 : R0 / 2 - 4, re = R1 / 2 + 4 > hh ? hh : R1 / 2 + 4;
    rb &= ~1;
    re = (re + 1) & ~1;
    int a2 = rb / 2 - 4, b2 = re / 2 + 4;
    if (a2 < 0) a2 = 0;
    if (b2 > hh2) b2 = hh2;
    const int rowsets[4][2] = {{a2, b2}, {hh2 + a2, hh2 + b2}, {a1, b1}, {hh + a1, hh + b1}};
    int n = 0;
    for (int i = 0; i < 4; ++i) {
        if (rowsets[i][0] >= rowsets[i][1]) continue;
        out[n][0] = (uint32_t)rowsets[i][0] * (uint32_t)pcol;
        out[n][1] = (uint32_t)rowsets[i][1] * (uint32_t)pcol;
        ++n;
    }
    return n;
}

// Workgroups of the chip that an inverse launch is planned for.
#ifndef KHG_INV_SLOTS
#define KHG_INV_SLOTS 1024
#endif
#ifndef KHG_INV_SLOTS_L2
#define KHG_INV_SLOTS_L2 512  // level 2: a quarter of the samples; measured 26.8 us against 28.5 us with 1024 (8K frame)
#endif

// Plane of colour component comp at level 2 (into the LL buffer) or level 1 (into dst: the caller's f64 plane, or
// none where the launch converts to RGB) of an inverse launch, without its grid.
// Tiles: every rank decodes the whole symbol planes, so a tile only restricts the rows it reconstructs: its band at
// level 1, and at level 2 that band of the LL quadrant widened by 4 rows (level 1 reads two row pairs beyond).
static inline void khg_inv_plane(KhgInvPlane* pl, const KhgWorkspace& ws, const KhgDecWorkspace& dws, int rows, int cols,
                                 int comp, int level, KhgTile tile, double* dst)
{
    memset(pl, 0, sizeof *pl);
    const int prow = comp ? rows / 2 : rows, pcol = comp ? cols / 2 : cols;
    pl->sym = ws.sym[comp];
    pl->sym_pitch = pcol;
    pl->ll = ws.ll[comp];
    pl->ll_pitch = pcol / 2;
    if (level == 2) {
        pl->ph = prow / 2; pl->pw = pcol / 2;
        pl->dst = ws.ll[comp];
        pl->dst_pitch = pcol / 2;
    } else {
        pl->ph = prow; pl->pw = pcol;
        pl->dst = dst;
        pl->dst_pitch = pcol;
    }
    for (int lvl = 0; lvl < 3; ++lvl) khg_make_quant(rows, cols, lvl, comp != 0, &pl->q[lvl]);
    const int R0 = khg_tile_row0(prow, tile), R1 = khg_tile_row1(prow, tile);
    if (level == 1) {
        pl->row_begin = R0;
        pl->row_end = R1;
    } else {
        pl->row_begin = R0 / 2 - 4 < 0 ? 0 : R0 / 2 - 4;
        pl->row_end = R1 / 2 + 4 > pl->ph ? pl->ph : R1 / 2 + 4;
        pl->row_begin &= ~1;
        pl->row_end = (pl->row_end + 1) & ~1;
    }
}

// An inverse launch over planes comp0 .. comp0 + n - 1 at `level` (KHG_INV_PLANES2 / KHG_INV_PLANES1; dst: the
// outputs of the level-1 planes), or the grid of the fused launch (khg_inv_args_fused).
static inline KhgInvArgs khg_inv_args(const KhgWorkspace& ws, const KhgDecWorkspace& dws, int rows, int cols, int comp0,
                                      int n, int level, KhgTile tile = KhgTile{0, 1}, double* const dst[3] = nullptr)
{
    KhgInvArgs a;
    memset(&a, 0, sizeof a);
    a.hdr = dws.hdr;
    a.out_row0 = khg_tile_row0(rows, tile);
    a.nplanes = n;
    KhgGridGroup g[3];
    for (int i = 0; i < n; ++i) {
        khg_inv_plane(&a.pl[i], ws, dws, rows, cols, comp0 + i, level, tile, dst ? dst[comp0 + i] : nullptr);
        g[i].pw = a.pl[i].pw; g[i].span = a.pl[i].row_end - a.pl[i].row_begin; g[i].weight = 1;
        g[i].strip_pairs = KHG_STRIP_PAIRS;
    }
    a.nwg = khg_plan_groups(g, n, level == 2 ? KHG_INV_SLOTS_L2 : KHG_INV_SLOTS);
    for (int i = 0; i < n; ++i) {
        a.pl[i].seg_rows = g[i].seg_rows; a.pl[i].nseg = g[i].nseg; a.pl[i].nstrip_wg = g[i].nstrip_wg; a.pl[i].wg0 = g[i].wg0;
    }
    return a;
}

// The fused level-1 launch (khg_inv_fused_wg_body): the grid of the luma plane; a.pl[1] and a.pl[2] describe the
// chroma planes the same workgroups reconstruct.  r / g / b are set by the caller.
static inline KhgInvArgs khg_inv_args_fused(const KhgWorkspace& ws, const KhgDecWorkspace& dws, int rows, int cols,
                                            KhgTile tile = KhgTile{0, 1})
{
    KhgInvArgs a = khg_inv_args(ws, dws, rows, cols, 0, 1, 1, tile);
    khg_inv_plane(&a.pl[1], ws, dws, rows, cols, 1, 1, tile, nullptr);
    khg_inv_plane(&a.pl[2], ws, dws, rows, cols, 2, 1, tile, nullptr);
    return a;
}
