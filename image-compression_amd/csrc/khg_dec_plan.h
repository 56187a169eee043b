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

// Grid of one inverse-transform launch over a ph x pw plane.
static inline KhgFwdGrid khg_inv_grid(int ph, int pw)
{
    return khg_strip_grid(ph, (pw + KHG_ISTRIP - 1) / KHG_ISTRIP, 1280, 8, 512);  // 5 workgroups per CU
}

// Inverse-transform arguments of colour component comp at level 2 (into the LL buffer) or
// level 1 (into the component's f64 plane).
// Tiles: every rank decodes the whole symbol planes, so a tile only restricts the rows it reconstructs: its band at
// level 1, and at level 2 that band of the LL quadrant widened by 4 rows (level 1 reads two row pairs beyond).
static inline KhgInvArgs khg_inv_args(const KhgWorkspace& ws, const KhgDecWorkspace& dws, int rows, int cols,
                                      int comp, int level, KhgFwdGrid* grid, KhgTile tile = KhgTile{0, 1})
{
    KhgInvArgs a;
    memset(&a, 0, sizeof a);
    const int prow = comp ? rows / 2 : rows, pcol = comp ? cols / 2 : cols;
    a.sym = ws.sym[comp];
    a.sym_pitch = pcol;
    a.ll = ws.ll[comp];
    a.ll_pitch = pcol / 2;
    if (level == 2) {
        a.ph = prow / 2; a.pw = pcol / 2;
        a.dst = ws.ll[comp];
        a.dst_pitch = pcol / 2;
    } else {
        a.ph = prow; a.pw = pcol;
        a.dst = dws.plane[comp];  // null for luma: its level-1 pass writes RGB instead
        a.dst_pitch = pcol;
    }
    a.cb = dws.plane[1];
    a.cr = dws.plane[2];
    a.hdr = dws.hdr;
    for (int lvl = 0; lvl < 3; ++lvl) khg_make_quant(rows, cols, lvl, comp != 0, &a.q[lvl]);
    const int R0 = khg_tile_row0(prow, tile), R1 = khg_tile_row1(prow, tile);
    if (level == 1) {
        a.row_begin = R0;
        a.row_end = R1;
    } else {
        a.row_begin = R0 / 2 - 4 < 0 ? 0 : R0 / 2 - 4;
        a.row_end = R1 / 2 + 4 > a.ph ? a.ph : R1 / 2 + 4;
    }
    a.out_row0 = khg_tile_row0(rows, tile);
    *grid = khg_inv_grid(a.row_end - a.row_begin, a.pw);
    a.seg_rows = grid->seg_rows;
    a.nstrip = grid->nstrip;
    return a;
}
