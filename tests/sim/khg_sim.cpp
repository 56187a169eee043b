// khg_sim.cpp -- TEST-ONLY sequential simulation of the HIP kernel bodies.
//
// Compiles image-compression_amd/csrc/*.h with -DKHG_SIM: every workgroup body runs its
// threads one after another inside each barrier-separated phase, on host memory.  This
// checks indexing, arithmetic order and the launch plan on a machine without a GPU; it
// is never linked into, loaded by or used as a fallback for the product library.
#include <stdlib.h>
#include <string.h>

#include <vector>

#define KHG_SIM 1
#include "khg_plan.h"
#include "khg_dec_plan.h"
#include "khg_synth.h"
#include "khg_visual.h"
#include "khg_index.h"

#ifndef KHG_SIM_SLACK
#define KHG_SIM_SLACK 64  // bytes past each workspace array; 0 under AddressSanitizer to expose any overrun
#endif

namespace {

struct SimWs {
    KhgWorkspace ws;
    std::vector<std::vector<char>> bufs;
    template <class T> T* alloc(size_t bytes) {
        bufs.emplace_back(bytes + KHG_SIM_SLACK, 0);
        return reinterpret_cast<T*>(bufs.back().data());
    }
    SimWs(int rows, int cols) {
        KhgWorkspaceSizes s;
        khg_workspace_sizes(rows, cols, &s);
        for (int p = 0; p < 3; ++p) { ws.ll[p] = alloc<double>(s.ll[p]); ws.sym[p] = alloc<int16_t>(s.sym[p]); }
        ws.hist_cnt = alloc<uint32_t>(s.hist); ws.hist_first = alloc<uint32_t>(s.hist);
        ws.tab_cnt = alloc<int32_t>(s.tab); ws.tab_sym = alloc<int32_t>(s.tab); ws.tab_k = alloc<int32_t>(16);
        ws.codetab = alloc<uint64_t>(s.codetab); ws.big_scratch = alloc<uint32_t>(s.big_scratch);
        ws.tree_left = alloc<int32_t>(s.tree); ws.tree_right = alloc<int32_t>(s.tree); ws.tree_root = alloc<int32_t>(16);
        ws.chunk_bits = alloc<uint32_t>(s.chunk_bits); ws.chunk_off = alloc<uint64_t>(s.chunk_off);
        ws.chain = alloc<uint64_t>(s.chain);
        ws.res = alloc<KhgResult>(sizeof(KhgResult));
        KhgDecWorkspaceSizes d;
        khg_dec_workspace_sizes(rows, cols, &d);
        dws.lut = alloc<uint32_t>(d.lut);
        dws.lut2 = alloc<uint32_t>(d.lut2);
        dws.blk = alloc<KhgBlkInfo>(d.blk);
        dws.walk = alloc<uint32_t>(d.walk);
        dws.hdr = alloc<KhgDecHeader>(sizeof(KhgDecHeader));
    }
    KhgDecWorkspace dws;
};

template <int KIND>
void run_fwd(const KhgFwdArgs& a) {
    KhgFwdSmemAny* sh = new KhgFwdSmemAny;
    for (int wg = 0; wg < a.nwg; ++wg) {
        if (a.lp == 2) khg_fwd_launch_body<KIND, 2, KHG_FWD_LPC, true>(a, wg, *sh);
        else khg_fwd_launch_body<KIND, 1, 1, true>(a, wg, *sh);
    }
    delete sh;
}

void forward_all(SimWs& S, const uint8_t* r, const uint8_t* g, const uint8_t* b, int rows, int cols, int lvl,
                 double* const coef[3], KhgTile tile = KhgTile{0, 1}, int bgr_stride = 0) {
    KhgWorkspace& ws = S.ws;
    memset(ws.hist_cnt, 0, 3 * KHG_NBINS * 4);
    memset(ws.hist_first, 0xFF, 3 * KHG_NBINS * 4);
    memset(ws.res, 0, sizeof(KhgResult));
    if (bgr_stride) run_fwd<KHG_FWD_BGR>(khg_fwd_args_rgb(ws, r, g, b, rows, cols, lvl, coef, tile, false, bgr_stride));
    else run_fwd<KHG_FWD_RGB>(khg_fwd_args_rgb(ws, r, g, b, rows, cols, lvl, coef, tile));
    run_fwd<KHG_FWD_PLANES2>(khg_fwd_args_planes(ws, rows, cols, lvl, 2, coef, tile));
}

}  // namespace

extern "C" {

int khgsim_forward(const uint8_t* r, const uint8_t* g, const uint8_t* b, int rows, int cols, int lvl,
                   double* y, double* cb, double* cr, int16_t* qy, int16_t* qcb, int16_t* qcr) {
    if (!khg_dims_ok(rows, cols)) return -1;
    SimWs S(rows, cols);
    double* coef[3] = {y, cb, cr};
    forward_all(S, r, g, b, rows, cols, lvl, coef);
    size_t ny = (size_t)rows * cols;
    if (qy) memcpy(qy, S.ws.sym[0], ny * 2);
    if (qcb) memcpy(qcb, S.ws.sym[1], ny / 4 * 2);
    if (qcr) memcpy(qcr, S.ws.sym[2], ny / 4 * 2);
    return S.ws.res->status;
}

// khg_packscan_kernel + khg_meta_kernel: the entries of the chain one after another (tickets 0, 1, 2, ...), on state
// words filled with junk first (the job-init launch zeroes them on the GPU; here only what the bodies rely on).
static void sim_pack_single_pass(const KhgEncArgs& e) {
    const uint32_t entries = e.chain_entries, groups = (entries + KHG_CHAIN_GROUP - 1) / KHG_CHAIN_GROUP;
    memset(e.chain, 0, (KHG_CHAIN_HEAD + 2 * ((size_t)entries + groups)) * sizeof(uint64_t));
    KhgPackScanSmem* psh = new KhgPackScanSmem;
    for (uint32_t t = 0; t < entries; ++t) khg_packscan_body(e, t, *psh);
    delete psh;
    KhgScanSmem* ssh = new KhgScanSmem;
    khg_meta_body(e, *ssh);
    delete ssh;
}

// format 2 (khg_ctx_set_format): the seek index behind the file
static void sim_append_index(const KhgEncArgs& e) {
    KhgIdxSmem* ish = new KhgIdxSmem;
    for (uint32_t t = 0; t < e.chain_entries; ++t) khg_index_body(e, t, *ish);
    delete ish;
    khg_index_meta_body(e);
}

static int sim_encode(SimWs& S, const uint8_t* r, const uint8_t* g, const uint8_t* b, int rows, int cols, int lvl,
                      uint8_t* out, size_t cap, size_t* out_len, int bgr_stride = 0, int format = 1) {
    forward_all(S, r, g, b, rows, cols, lvl, nullptr, KhgTile{0, 1}, bgr_stride);
    KhgTableArgs t = khg_table_args(S.ws);
    KhgTableSmem* tsh = new KhgTableSmem;
    for (int p = 0; p < 3; ++p) khg_table_body(t, p, *tsh);
    delete tsh;
    KhgEncArgs e = khg_enc_args(S.ws, rows, cols, lvl, out, cap);
    sim_pack_single_pass(e);
    if (format == 2) sim_append_index(e);
    if (out_len) *out_len = (size_t)S.ws.res->total_len;
    return S.ws.res->status;
}

// The entropy coder alone, on symbol planes given by the caller (tests craft code tables no picture would give):
// histogram and first raster indices as the forward kernels leave them, then the table, packer and header kernels.
int khgsim_pack_symbols(const int16_t* qy, const int16_t* qcb, const int16_t* qcr, int rows, int cols, int lvl,
                        uint8_t* out, size_t cap, size_t* out_len) {
    if (!khg_dims_ok(rows, cols) || lvl < 0 || lvl > 2) return -1;
    SimWs S(rows, cols);
    KhgWorkspace& ws = S.ws;
    memset(ws.hist_cnt, 0, 3 * KHG_NBINS * 4);
    memset(ws.hist_first, 0xFF, 3 * KHG_NBINS * 4);
    memset(ws.res, 0, sizeof(KhgResult));
    const int16_t* q[3] = {qy, qcb, qcr};
    for (int p = 0; p < 3; ++p) {
        const size_t n = (size_t)rows * cols / (p ? 4 : 1);
        memcpy(ws.sym[p], q[p], n * 2);
        for (size_t i = 0; i < n; ++i) {
            const int v = q[p][i];
            if (v < -KHG_SYM_BIAS || v >= KHG_SYM_BIAS) return -1;
            const size_t bin = (size_t)p * KHG_NBINS + (size_t)(v + KHG_SYM_BIAS);
            if (ws.hist_cnt[bin]++ == 0) ws.hist_first[bin] = (uint32_t)i;
        }
    }
    KhgTableArgs t = khg_table_args(ws);
    KhgTableSmem* tsh = new KhgTableSmem;
    for (int p = 0; p < 3; ++p) khg_table_body(t, p, *tsh);
    delete tsh;
    KhgEncArgs e = khg_enc_args(ws, rows, cols, lvl, out, cap);
    sim_pack_single_pass(e);
    if (out_len) *out_len = (size_t)ws.res->total_len;
    return ws.res->status;
}

int khgsim_encode_indexed(const uint8_t* r, const uint8_t* g, const uint8_t* b, int rows, int cols, int lvl,
                          uint8_t* out, size_t cap, size_t* out_len) {
    if (!khg_dims_ok(rows, cols) || lvl < 0 || lvl > 2) return -1;
    SimWs S(rows, cols);
    return sim_encode(S, r, g, b, rows, cols, lvl, out, cap, out_len, 0, 2);
}

// the pixel array of a 24-bit BMP (B, G, R per pixel, rows bottom-up, `stride` bytes per row)
int khgsim_encode_bgr(const uint8_t* pixels, size_t stride, int rows, int cols, int lvl, uint8_t* out, size_t cap,
                      size_t* out_len) {
    if (!khg_dims_ok(rows, cols) || lvl < 0 || lvl > 2 || stride < 3 * (size_t)cols) return -1;
    SimWs S(rows, cols);
    return sim_encode(S, pixels, nullptr, nullptr, rows, cols, lvl, out, cap, out_len, (int)stride);
}

int khgsim_encode(const uint8_t* r, const uint8_t* g, const uint8_t* b, int rows, int cols, int lvl,
                  uint8_t* out, size_t cap, size_t* out_len) {
    if (!khg_dims_ok(rows, cols) || lvl < 0 || lvl > 2) return -1;
    SimWs S(rows, cols);
    return sim_encode(S, r, g, b, rows, cols, lvl, out, cap, out_len);
}

// The chain over the workgroup maps: in one piece (short streams on the GPU) or, when KHGSIM_CHAIN_SPLIT is set in
// the environment, as the two launches long streams take (batches of 16 * KHG_CHAIN_GROUPS maps).
static void sim_chain(const KhgDecArgs& d, KhgDecChainSmem& one) {
    const char* env = getenv("KHGSIM_CHAIN_SPLIT");
    if (!env || !*env || *env == '0') {
        for (int p = 0; p < 3; ++p) khg_dec_chain_body(d, p, one);
        return;
    }
    KhgDecChainSplitSmem* sh = new KhgDecChainSplitSmem;
    const uint32_t nbatch = (uint32_t)(((8 * d.len + KHG_BLK_BITS - 1) / KHG_BLK_BITS + 1 + KHG_CHAIN_BATCH - 1) / KHG_CHAIN_BATCH);
    for (int p = 0; p < 3; ++p)
        for (uint32_t b = 0; b < nbatch; ++b) khg_dec_chain_maps_body(d, p, b, *sh);
    for (int p = 0; p < 3; ++p)
        for (uint32_t b = nbatch; b-- > 0;) khg_dec_chain_finish_body(d, p, b, *sh);  // any order
    delete sh;
}

// Mirrors khg_decode_dev / khg_tile_decode_dev (r, g, b receive the tile's rows only) / khg_decode_ycbcr_dev.
static void sim_inverse(SimWs& S, int rows, int cols, uint8_t* r, uint8_t* g, uint8_t* b, KhgTile tile,
                        double* const ycbcr[3], size_t bgr_stride = 0) {
    KhgInvSmemAny* ish = new KhgInvSmemAny;
    {
        const KhgInvArgs a = khg_inv_args(S.ws, S.dws, rows, cols, 0, 3, 2, tile);
        for (int wg = 0; wg < a.nwg; ++wg) khg_inv_launch_body<KHG_INV_PLANES2>(a, wg, *ish);
    }
    if (ycbcr) {
        const KhgInvArgs a = khg_inv_args(S.ws, S.dws, rows, cols, 0, 3, 1, tile, ycbcr);
        for (int wg = 0; wg < a.nwg; ++wg) khg_inv_launch_body<KHG_INV_PLANES1>(a, wg, *ish);
    } else {
        KhgInvArgs ay = khg_inv_args_fused(S.ws, S.dws, rows, cols, tile);
        ay.r = r; ay.g = g; ay.b = b;
        ay.bgr_stride = bgr_stride;
        for (int wg = 0; wg < ay.nwg; ++wg) {
            if (bgr_stride) khg_inv_launch_body<KHG_INV_BGR_FUSED>(ay, wg, *ish);
            else khg_inv_launch_body<KHG_INV_RGB_FUSED>(ay, wg, *ish);
        }
    }
    delete ish;
}

static int sim_decode(SimWs& S, const uint8_t* comp, size_t len, int rows, int cols, uint8_t* r, uint8_t* g, uint8_t* b,
                      int16_t* qy, int16_t* qcb, int16_t* qcr, int* passes_used, KhgTile tile,
                      double* const ycbcr[3] = nullptr, size_t bgr_stride = 0, bool indexed = false) {
    std::vector<uint8_t> buf((len + 3) & ~(size_t)3, 0);
    memcpy(buf.data(), comp, len);
    KhgDecArgs d = khg_dec_args(S.ws, S.dws, buf.data(), len, rows, cols);
    khg_dec_header_body(d);  // clears the result block itself
    KhgDecTreeSmem* tsh = new KhgDecTreeSmem;
    for (int p = 0; p < 3; ++p) khg_dec_tree_body(d, p, *tsh);
    delete tsh;
    KhgDecSmem<KHG_MAP_SMALL>* lsh = new KhgDecSmem<KHG_MAP_SMALL>;
    KhgDecSmem<KHG_MAXOFF>* lsh2 = new KhgDecSmem<KHG_MAXOFF>;
    KhgDecWriteSmem* wsh = new KhgDecWriteSmem;
    KhgDecChainSmem* ssh = new KhgDecChainSmem;
    if (indexed) {
        for (uint32_t b_ = 0; b_ < d.nblocks + 3; ++b_) khg_dec_index_body(d, b_);
    } else {
        for (uint32_t b_ = 0; b_ < d.nblocks; ++b_) khg_dec_map_body<KHG_MAP_SMALL>(d, b_, *lsh);
        for (uint32_t b_ = 0; b_ < d.nblocks; ++b_) khg_dec_map_body<KHG_MAXOFF>(d, b_, *lsh2);
        sim_chain(d, *ssh);
    }
    if (passes_used) *passes_used = 1;
    for (uint32_t b_ = 0; b_ < d.nblocks; ++b_) khg_dec_write_body(d, b_, *wsh);
    delete lsh;
    delete lsh2;
    delete wsh;
    delete ssh;
    sim_inverse(S, rows, cols, r, g, b, tile, ycbcr, bgr_stride);
    size_t ny = (size_t)rows * cols;
    if (qy) memcpy(qy, S.ws.sym[0], ny * 2);
    if (qcb) memcpy(qcb, S.ws.sym[1], ny / 4 * 2);
    if (qcr) memcpy(qcr, S.ws.sym[2], ny / 4 * 2);
    return S.ws.res->status;
}

int khgsim_decode(const uint8_t* comp, size_t len, int rows, int cols, uint8_t* r, uint8_t* g, uint8_t* b,
                  int16_t* qy, int16_t* qcb, int16_t* qcr, int* passes_used) {
    if (!khg_dims_ok(rows, cols)) return -1;
    SimWs S(rows, cols);
    return sim_decode(S, comp, len, rows, cols, r, g, b, qy, qcb, qcr, passes_used, KhgTile{0, 1});
}

// Mirrors khg_decode_indexed_dev.  poison: the workspace is filled with it first (what another job left behind).
int khgsim_decode_indexed(const uint8_t* comp, size_t len, int rows, int cols, uint8_t* r, uint8_t* g, uint8_t* b,
                          int poison) {
    if (!khg_dims_ok(rows, cols)) return -1;
    SimWs S(rows, cols);
    if (poison >= 0)
        for (auto& buf : S.bufs) memset(buf.data(), poison, buf.size() - KHG_SIM_SLACK);
    return sim_decode(S, comp, len, rows, cols, r, g, b, nullptr, nullptr, nullptr, nullptr, KhgTile{0, 1}, nullptr, 0,
                      true);
}

// Mirrors khg_decode_bgr_dev: the picture as a BMP pixel array of `stride` bytes per row.
int khgsim_decode_bgr(const uint8_t* comp, size_t len, int rows, int cols, uint8_t* pixels, size_t stride) {
    if (!khg_dims_ok(rows, cols) || stride < 3 * (size_t)cols || (stride & 1)) return -1;
    SimWs S(rows, cols);
    return sim_decode(S, comp, len, rows, cols, pixels, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                      KhgTile{0, 1}, nullptr, stride);
}

// ---- the by-product pictures (khg_vis_*_dev)
static void sim_vis_ycc(KhgVisYccArgs& v) {
    const uint32_t nb = (uint32_t)(((size_t)v.rows * v.cols + 255) / 256);
    for (uint32_t b_ = 0; b_ < nb; ++b_) khg_vis_ycc_body(v, b_);
}
int khgsim_vis_coeffs(const uint8_t* r, const uint8_t* g, const uint8_t* b, int rows, int cols, uint8_t* orr, uint8_t* og,
                      uint8_t* ob) {
    if (!khg_dims_ok(rows, cols)) return -1;
    SimWs S(rows, cols);
    const size_t n = (size_t)rows * cols;
    std::vector<double> planes(n + n / 2);
    double* coef[3] = {planes.data(), planes.data() + n, planes.data() + n + n / 4};
    forward_all(S, r, g, b, rows, cols, 0, coef);
    KhgVisYccArgs v;
    memset(&v, 0, sizeof v);
    v.rows = rows; v.cols = cols;
    v.y = coef[0]; v.cb = coef[1]; v.cr = coef[2];
    v.r = orr; v.g = og; v.b = ob;
    sim_vis_ycc(v);
    return 0;
}
int khgsim_vis_dequant(const uint8_t* comp, size_t len, int rows, int cols, uint8_t* orr, uint8_t* og, uint8_t* ob) {
    if (!khg_dims_ok(rows, cols)) return -1;
    SimWs S(rows, cols);
    std::vector<uint8_t> scratch(3 * (size_t)rows * cols);
    int rc = sim_decode(S, comp, len, rows, cols, scratch.data(), scratch.data() + (size_t)rows * cols,
                        scratch.data() + 2 * (size_t)rows * cols, nullptr, nullptr, nullptr, nullptr, KhgTile{0, 1});
    if (rc) return rc;
    KhgVisYccArgs v;
    memset(&v, 0, sizeof v);
    v.rows = rows; v.cols = cols;
    for (int p = 0; p < 3; ++p) v.sym[p] = S.ws.sym[p];
    v.hdr = S.dws.hdr;
    for (int lvl = 0; lvl < 3; ++lvl) {
        khg_make_quant(rows, cols, lvl, 0, &v.q[0][lvl]);
        khg_make_quant(rows, cols, lvl, 1, &v.q[1][lvl]);
    }
    v.r = orr; v.g = og; v.b = ob;
    sim_vis_ycc(v);
    return 0;
}
int khgsim_vis_steps(int rows, int cols, int lvl, uint8_t* r, uint8_t* g, uint8_t* b, double* lo_hi) {
    if (!khg_dims_ok(rows, cols) || lvl < 0 || lvl > 2) return -1;
    KhgVisStepArgs a;
    a.rows = rows; a.cols = cols; a.lvl = lvl;
    khg_make_quant(rows, cols, lvl, 0, &a.q);
    khg_vis_step_range(rows, cols, lvl, a.q, &a.lo, &a.hi);
    a.r = r; a.g = g; a.b = b;
    const uint32_t nb = (uint32_t)(((size_t)rows * cols + 255) / 256);
    for (uint32_t b_ = 0; b_ < nb; ++b_) khg_vis_step_body(a, b_);
    if (lo_hi) { lo_hi[0] = a.lo; lo_hi[1] = a.hi; }
    return 0;
}
int khgsim_vis_bars(int rows, int cols, const long long* len, uint8_t* r, uint8_t* g, uint8_t* b) {
    if (!khg_dims_ok(rows, cols)) return -1;
    KhgVisBarArgs a;
    a.rows = rows; a.cols = cols;
    for (int c = 0; c < 3; ++c) a.len[c] = len[c];
    a.r = r; a.g = g; a.b = b;
    const uint32_t nb = (uint32_t)(((size_t)rows * cols + 255) / 256);
    for (uint32_t b_ = 0; b_ < nb; ++b_) khg_vis_bar_body(a, b_);
    return 0;
}

int khgsim_downsample(const uint8_t* src, int sh_, int sw, uint8_t* dst, int dh, int dw) {
    KhgDownArgs a;
    a.src = src; a.src_h = sh_; a.src_w = sw; a.dst = dst; a.dst_h = dh; a.dst_w = dw;
    const uint32_t nb = (uint32_t)(((size_t)dh * dw + 255) / 256);
    for (uint32_t b_ = 0; b_ < nb; ++b_) khg_downsample_body(a, b_);
    return 0;
}

// ---- one long-lived workspace, as a khg_ctx has: jobs of any size up to its maximum, one after another, and a
// poison fill that stands for whatever an earlier job left behind (khg_debug_poison)
struct SimCtx {
    SimWs S;
    int rows, cols;
    SimCtx(int r, int c) : S(r, c), rows(r), cols(c) {}
};
void* khgsim_ctx_create(int rows, int cols) { return khg_dims_ok(rows, cols) ? new SimCtx(rows, cols) : nullptr; }
void khgsim_ctx_destroy(void* h) { delete (SimCtx*)h; }
void khgsim_ctx_poison(void* h, int byte) {
    for (auto& b : ((SimCtx*)h)->S.bufs) memset(b.data(), byte, b.size() - KHG_SIM_SLACK);
}
int khgsim_ctx_encode(void* h, const uint8_t* r, const uint8_t* g, const uint8_t* b, int rows, int cols, int lvl,
                      uint8_t* out, size_t cap, size_t* out_len) {
    SimCtx* C = (SimCtx*)h;
    if (!khg_dims_ok(rows, cols) || rows > C->rows || cols > C->cols || lvl < 0 || lvl > 2) return -1;
    return sim_encode(C->S, r, g, b, rows, cols, lvl, out, cap, out_len);
}
int khgsim_ctx_decode(void* h, const uint8_t* comp, size_t len, int rows, int cols, uint8_t* r, uint8_t* g, uint8_t* b) {
    SimCtx* C = (SimCtx*)h;
    if (!khg_dims_ok(rows, cols) || rows > C->rows || cols > C->cols) return -1;
    return sim_decode(C->S, comp, len, rows, cols, r, g, b, nullptr, nullptr, nullptr, nullptr, KhgTile{0, 1});
}

// Mirrors khg_encode_ycbcr_dev / khg_decode_ycbcr_dev.
int khgsim_encode_ycbcr(const double* y, const double* cb, const double* cr, int rows, int cols, int lvl, uint8_t* out,
                        size_t cap, size_t* out_len, double* coef_y, double* coef_cb, double* coef_cr) {
    if (!khg_dims_ok(rows, cols) || lvl < 0 || lvl > 2) return -1;
    SimWs S(rows, cols);
    KhgWorkspace& ws = S.ws;
    memset(ws.hist_cnt, 0, 3 * KHG_NBINS * 4);
    memset(ws.hist_first, 0xFF, 3 * KHG_NBINS * 4);
    memset(ws.res, 0, sizeof(KhgResult));
    const double* src[3] = {y, cb, cr};
    double* coef[3] = {coef_y, coef_cb, coef_cr};
    run_fwd<KHG_FWD_PLANES1>(khg_fwd_args_planes(ws, rows, cols, lvl, 1, coef_y ? coef : nullptr, KhgTile{0, 1}, src));
    run_fwd<KHG_FWD_PLANES2>(khg_fwd_args_planes(ws, rows, cols, lvl, 2, coef_y ? coef : nullptr));
    KhgTableArgs t = khg_table_args(ws);
    KhgTableSmem* tsh = new KhgTableSmem;
    for (int p = 0; p < 3; ++p) khg_table_body(t, p, *tsh);
    delete tsh;
    KhgEncArgs e = khg_enc_args(ws, rows, cols, lvl, out, cap);
    sim_pack_single_pass(e);
    if (out_len) *out_len = (size_t)ws.res->total_len;
    return ws.res->status;
}

int khgsim_decode_ycbcr(const uint8_t* comp, size_t len, int rows, int cols, double* y, double* cb, double* cr) {
    if (!khg_dims_ok(rows, cols)) return -1;
    SimWs S(rows, cols);
    double* planes[3] = {y, cb, cr};
    return sim_decode(S, comp, len, rows, cols, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                      KhgTile{0, 1}, planes);
}

// ---- one rank of a row-tiled image: the steps of khg_tile_* on host memory
struct SimTile {
    SimWs S;
    KhgTile t;
    int rows, cols, lvl;
    // partitioned decode in progress
    std::vector<uint8_t> comp;
    size_t comp_len = 0;
    uint32_t nb[3] = {0, 0, 0};
    SimTile(int r, int c) : S(r, c), t{0, 1}, rows(r), cols(c), lvl(0) {}
    KhgDecArgs dec_args() {
        KhgDecArgs d = khg_dec_args(S.ws, S.dws, comp.data(), comp_len, rows, cols);
        d.part_rank = (uint32_t)t.rank;
        d.part_nranks = (uint32_t)t.nranks;
        return d;
    }
    uint32_t grid() const {
        uint32_t g = 0;
        for (int p = 0; p < 3; ++p)
            g += khg_part_begin(nb[p], (uint32_t)t.rank + 1, (uint32_t)t.nranks) - khg_part_begin(nb[p], (uint32_t)t.rank, (uint32_t)t.nranks);
        return g;
    }
};

void* khgsim_tile_create(int rows, int cols) { return khg_dims_ok(rows, cols) ? new SimTile(rows, cols) : nullptr; }
void khgsim_tile_destroy(void* h) { delete (SimTile*)h; }

size_t khgsim_tile_chunk_count(int rows, int cols, int nranks) {
    KhgWorkspace none;
    memset(&none, 0, sizeof none);
    return khg_enc_args(none, rows, cols, 0, nullptr, 0, KhgTile{0, nranks}).nchunks;
}

int khgsim_tile_forward(void* h, const uint8_t* r, const uint8_t* g, const uint8_t* b, int rows, int cols, int lvl,
                        int rank, int nranks) {
    SimTile* T = (SimTile*)h;
    KhgTile t = {rank, nranks};
    if (!khg_tile_ok(rows, t) || rows != T->rows || cols != T->cols) return -1;
    T->t = t;
    T->lvl = lvl;
    forward_all(T->S, r, g, b, rows, cols, lvl, nullptr, t);
    return T->S.ws.res->status;
}

void khgsim_tile_hist_get(void* h, uint32_t* cnt, uint32_t* first) {
    SimTile* T = (SimTile*)h;
    memcpy(cnt, T->S.ws.hist_cnt, 3 * KHG_NBINS * 4);
    memcpy(first, T->S.ws.hist_first, 3 * KHG_NBINS * 4);
}

void khgsim_tile_hist_set(void* h, const uint32_t* cnt, const uint32_t* first) {
    SimTile* T = (SimTile*)h;
    memcpy(T->S.ws.hist_cnt, cnt, 3 * KHG_NBINS * 4);
    memcpy(T->S.ws.hist_first, first, 3 * KHG_NBINS * 4);
}

void khgsim_tile_lens(void* h) {
    SimTile* T = (SimTile*)h;
    KhgTableArgs t = khg_table_args(T->S.ws);
    KhgTableSmem* tsh = new KhgTableSmem;
    for (int p = 0; p < 3; ++p) khg_table_body(t, p, *tsh);
    delete tsh;
    KhgEncArgs e = khg_enc_args(T->S.ws, T->rows, T->cols, T->lvl, nullptr, 0, T->t);
    memset(T->S.ws.chunk_bits, 0, (size_t)e.nchunks * 4);
    KhgLensSmem lsh;
    for (uint32_t c = 0; c < e.nchunks; ++c) khg_lens_body(e, c, lsh);
}

void khgsim_tile_bits_get(void* h, uint32_t* bits) {
    SimTile* T = (SimTile*)h;
    memcpy(bits, T->S.ws.chunk_bits, khgsim_tile_chunk_count(T->rows, T->cols, T->t.nranks) * 4);
}

void khgsim_tile_chunk_off_get(void* h, uint64_t* off) {
    SimTile* T = (SimTile*)h;
    memcpy(off, T->S.ws.chunk_off, khgsim_tile_chunk_count(T->rows, T->cols, T->t.nranks) * 8);
}

void khgsim_tile_bits_set(void* h, const uint32_t* bits) {
    SimTile* T = (SimTile*)h;
    memcpy(T->S.ws.chunk_bits, bits, khgsim_tile_chunk_count(T->rows, T->cols, T->t.nranks) * 4);
}

int khgsim_tile_pack(void* h, uint8_t* out, size_t cap, size_t* total) {
    SimTile* T = (SimTile*)h;
    KhgEncArgs e = khg_enc_args(T->S.ws, T->rows, T->cols, T->lvl, out, cap, T->t);
    // as khg_tile_pack_dev: offsets and length, the file's bytes zeroed (junk behind them), packer, header + tables
    memset(out, 0xA5, cap);
    KhgEncArgs scan = e;
    scan.write_meta = 0;
    KhgScanSmem* ssh = new KhgScanSmem;
    khg_scan_body(scan, *ssh);
    if (T->S.ws.res->ok_to_pack) {
        size_t bytes = ((size_t)T->S.ws.res->total_len + 15) & ~(size_t)15;
        if (bytes > (cap & ~(size_t)3)) bytes = cap & ~(size_t)3;
        memset(out, 0, bytes);
    }
    KhgPackSmem* psh = new KhgPackSmem;
    for (uint32_t c = 0; c < e.nchunks; ++c) khg_pack_body(e, c, *psh);
    delete psh;
    if (e.write_meta) khg_meta_body(e, *ssh);
    delete ssh;
    *total = (size_t)T->S.ws.res->total_len;
    return T->S.ws.res->status;
}

int khgsim_tile_decode(void* h, const uint8_t* comp, size_t len, int rows, int cols, int rank, int nranks, uint8_t* r,
                       uint8_t* g, uint8_t* b) {
    SimTile* T = (SimTile*)h;
    KhgTile t = {rank, nranks};
    if (!khg_tile_ok(rows, t) || rows != T->rows || cols != T->cols) return -1;
    return sim_decode(T->S, comp, len, rows, cols, r, g, b, nullptr, nullptr, nullptr, nullptr, t);
}

// ---- tiled decode with the entropy decode split over the ranks: the steps of khg_tile_decode_map_dev ..
// khg_tile_decode_inverse_dev on host memory.  The symbol planes are filled with a junk value first, so a symbol
// that is neither decoded here nor received from its owner shows in the picture.
int khgsim_tile_decode_map(void* h, const uint8_t* comp, size_t len, int rows, int cols, int rank, int nranks) {
    SimTile* T = (SimTile*)h;
    KhgTile t = {rank, nranks};
    if (!khg_tile_ok(rows, t) || rows != T->rows || cols != T->cols || len < 60) return -1;
    T->t = t;
    T->comp.assign((len + 3) & ~(size_t)3, 0);
    memcpy(T->comp.data(), comp, len);
    T->comp_len = len;
    khg_tile_active_blocks(T->comp.data(), T->nb);
    const size_t ny = (size_t)rows * cols;
    for (int p = 0; p < 3; ++p) memset(T->S.ws.sym[p], 0x7B, (p ? ny / 4 : ny) * 2);
    KhgDecArgs d = T->dec_args();
    for (int p = 0; p < 3; ++p) {
        const uint32_t cap = (p < 2 ? d.blk0[p + 1] : d.nblocks) - d.blk0[p];
        if (T->nb[p] > cap) T->nb[p] = cap;
        memset(T->S.dws.blk + d.blk0[p], 0, (size_t)T->nb[p] * sizeof(KhgBlkInfo));
    }
    khg_dec_header_body(d);
    KhgDecTreeSmem* tsh = new KhgDecTreeSmem;
    for (int p = 0; p < 3; ++p) khg_dec_tree_body(d, p, *tsh);
    delete tsh;
    KhgDecSmem<KHG_MAP_SMALL>* lsh = new KhgDecSmem<KHG_MAP_SMALL>;
    KhgDecSmem<KHG_MAXOFF>* lsh2 = new KhgDecSmem<KHG_MAXOFF>;
    const uint32_t grid = T->grid();
    for (uint32_t b_ = 0; b_ < grid; ++b_) {
        const uint32_t blk = khg_dec_dispatch_order(d, b_);
        if (blk == ~0u) continue;
        khg_dec_map_body<KHG_MAP_SMALL>(d, blk, *lsh);
        khg_dec_map_body<KHG_MAXOFF>(d, blk, *lsh2);
    }
    delete lsh;
    delete lsh2;
    return 0;
}

size_t khgsim_tile_blk_words(void* h) {
    SimTile* T = (SimTile*)h;
    return ((size_t)T->nb[0] + T->nb[1] + T->nb[2]) * (sizeof(KhgBlkInfo) / 4);
}

static void sim_blk_copy(SimTile* T, uint32_t* words, bool get) {
    KhgDecArgs d = T->dec_args();
    size_t at = 0;
    for (int p = 0; p < 3; ++p) {
        const size_t bytes = (size_t)T->nb[p] * sizeof(KhgBlkInfo);
        void* rec = T->S.dws.blk + d.blk0[p];
        if (get) memcpy(words + at, rec, bytes);
        else memcpy(rec, words + at, bytes);
        at += bytes / 4;
    }
}
void khgsim_tile_blk_get(void* h, uint32_t* words) { sim_blk_copy((SimTile*)h, words, true); }
void khgsim_tile_blk_set(void* h, const uint32_t* words) { sim_blk_copy((SimTile*)h, (uint32_t*)words, false); }

int khgsim_tile_decode_write(void* h, uint32_t* bounds) {
    SimTile* T = (SimTile*)h;
    KhgDecArgs d = T->dec_args();
    const int n = T->t.nranks;
    KhgDecChainSmem* ssh = new KhgDecChainSmem;
    sim_chain(d, *ssh);
    delete ssh;
    KhgDecWriteSmem* wsh = new KhgDecWriteSmem;
    const uint32_t grid = T->grid();
    for (uint32_t b_ = 0; b_ < grid; ++b_) {
        const uint32_t blk = khg_dec_dispatch_order(d, b_);
        if (blk != ~0u) khg_dec_write_body(d, blk, *wsh);
    }
    delete wsh;
    for (int p = 0; p < 3; ++p) {
        uint32_t total = (uint32_t)T->S.ws.res->bits[p];
        if (total > d.n[p]) total = d.n[p];
        bounds[p * (n + 1) + n] = total;
        for (int t = n - 1; t >= 0; --t) {
            const uint32_t b = khg_part_begin(T->nb[p], (uint32_t)t, (uint32_t)n);
            uint32_t v = b < T->nb[p] ? T->S.dws.blk[d.blk0[p] + b].out : 0xFFFFFFFFu;
            if (v > bounds[p * (n + 1) + t + 1]) v = bounds[p * (n + 1) + t + 1];
            bounds[p * (n + 1) + t] = v;
        }
    }
    return T->S.ws.res->status;
}

int khgsim_tile_need_ranges(int rows, int cols, int rank, int nranks, int plane, uint32_t* ranges) {
    KhgTile t = {rank, nranks};
    uint32_t r[4][2];
    const int n = khg_tile_need_ranges_impl(rows, cols, t, plane, r);
    for (int i = 0; i < n; ++i) { ranges[2 * i] = r[i][0]; ranges[2 * i + 1] = r[i][1]; }
    return n;
}

void khgsim_tile_sym_get(void* h, int plane, uint32_t first, uint32_t count, int16_t* dst) {
    memcpy(dst, ((SimTile*)h)->S.ws.sym[plane] + first, (size_t)count * 2);
}
void khgsim_tile_sym_set(void* h, int plane, uint32_t first, uint32_t count, const int16_t* src) {
    memcpy(((SimTile*)h)->S.ws.sym[plane] + first, src, (size_t)count * 2);
}

int khgsim_tile_decode_inverse(void* h, uint8_t* r, uint8_t* g, uint8_t* b) {
    SimTile* T = (SimTile*)h;
    sim_inverse(T->S, T->rows, T->cols, r, g, b, T->t, nullptr);
    return T->S.ws.res->status;
}

double khgsim_ssim(const uint8_t* a, const uint8_t* b, int width, int height, double* blocks_out) {
    size_t nblk = (size_t)(width / 8) * (height / 8);
    std::vector<double> blocks(nblk);
    KhgSsimArgs s;
    s.a = a; s.b = b; s.width = width; s.height = height; s.blocks = blocks.data();
    for (uint32_t bid = 0; bid < (nblk + 255) / 256; ++bid) khg_ssim_body(s, bid);
    if (blocks_out) memcpy(blocks_out, blocks.data(), nblk * sizeof(double));
    double mean = 0.0;
    KhgSsimMeanArgs m;
    m.blocks = blocks.data(); m.nblk = nblk; m.out = &mean; m.stat = nullptr;
    KhgSsimMeanSmem* msh = new KhgSsimMeanSmem;
    khg_ssim_mean_body(m, *msh);
    delete msh;
    return mean;
}

// The mean kernel on any values: x[0] + x[1] + ... in that order, divided by n.  stat (optional, 3 counters): windows
// summed as integers, windows scanned, values added one at a time.
double khgsim_ordered_mean(const double* x, size_t n, unsigned long long* stat) {
    double mean = 0.0;
    if (stat) stat[0] = stat[1] = stat[2] = 0;
    KhgSsimMeanArgs m;
    m.blocks = x; m.nblk = n; m.out = &mean; m.stat = stat;
    KhgSsimMeanSmem* msh = new KhgSsimMeanSmem;
    khg_ssim_mean_body(m, *msh);
    delete msh;
    return mean;
}

void khgsim_sse(const uint8_t* const in[3], const uint8_t* const out[3], size_t n, unsigned long long sse[3]) {
    KhgSseArgs a;
    for (int c = 0; c < 3; ++c) { a.in[c] = in[c]; a.out[c] = out[c]; sse[c] = 0; }
    a.n = n; a.sse = sse;
    KhgSseSmem sh;
    for (uint32_t bid = 0; bid < 3; ++bid) khg_sse_body(a, bid, 3, sh);
}

void khgsim_synth(uint8_t* r, uint8_t* g, uint8_t* b, int rows, int cols, uint32_t seed, uint32_t index) {
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            size_t i = (size_t)y * cols + x;
            khg_synth_pixel((uint32_t)y, (uint32_t)x, (uint32_t)cols, seed, index, &r[i], &g[i], &b[i]);
        }
}

// Arithmetic shortcuts against the operations they replace: out_div[i] = khg_div_const(x[i], c, 1/c),
// out_round[i] = the quantiser's rounding of x[i] / d[i]; fast[i] = 1 when no division was needed.
// host-side planning: segment height and count chosen for a transform of ph rows
// The grid planner of the transform launches: groups of (pw, span, weight) -> per group seg_rows, nseg, nstrip_wg,
// wg0 (4 ints each, in `out`); returns the workgroup count.
int khgsim_plan_groups(const int* groups, int ngroup, int slots, int* out)
{
    KhgGridGroup g[3];
    for (int i = 0; i < ngroup; ++i) { g[i].pw = groups[3 * i]; g[i].span = groups[3 * i + 1]; g[i].weight = groups[3 * i + 2]; g[i].strip_pairs = KHG_STRIP_PAIRS; }
    const int n = khg_plan_groups(g, ngroup, slots);
    for (int i = 0; i < ngroup; ++i) { out[4 * i] = g[i].seg_rows; out[4 * i + 1] = g[i].nseg; out[4 * i + 2] = g[i].nstrip_wg; out[4 * i + 3] = g[i].wg0; }
    return n;
}

// Rows of segment `seg` (khg_segment_rows).
void khgsim_segment_rows(int begin, int end, int nseg, int seg, int* r0, int* r1) { khg_segment_rows(begin, end, nseg, seg, r0, r1); }

void khgsim_div_const(const double* x, size_t n, double c, double* out_div)
{
    const double rc = 1 / c;
    for (size_t i = 0; i < n; ++i) out_div[i] = khg_div_const(x[i], c, rc);
}

void khgsim_round_quot(const double* x, const double* d, size_t n, double* out_round, uint8_t* fast)
{
    for (size_t i = 0; i < n; ++i) {
        int q;
        uint32_t nhi;
        const bool ok = khg_quant_try(x[i], 1.0 / d[i], &q, &nhi);  // the kernels' shortcut; they divide when it declines
        out_round[i] = ok ? (double)q : round(x[i] / d[i]);
        fast[i] = ok;
    }
}

}  // extern "C"
