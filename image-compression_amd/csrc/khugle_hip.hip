// khugle_hip.hip -- gfx950 kernels (wrappers around the phase bodies) and the C ABI of
// include/khugle_hip.h.  Build: see image-compression_amd/csrc/Makefile (-ffp-contract=off
// is mandatory: the codec's f64 arithmetic must round every product and sum separately).
#include <hip/hip_runtime.h>

#include <new>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "khugle_hip.h"
#include "khg_plan.h"
#include "khg_synth.h"
#include "khg_dec_plan.h"
#include "khg_visual.h"
#include "khg_index.h"

// ------------------------------------------------------------------ kernels

// KIND: KHG_FWD_RGB (level 1 of Y, Cb, Cr from u8 RGB), KHG_FWD_PLANES1 (level 1 from f64 planes),
// KHG_FWD_PLANES2 (level 2 of the three components).
// LP: sample pairs per lane; with two the kernel is planned for two workgroups per CU (up to 256 registers).
// LPC: the same for the workgroups that carry Cb and Cr.
#ifndef KHG_FWD_LP2_WAVES
#define KHG_FWD_LP2_WAVES 4  // workgroups per CU (waves per SIMD) the two-pair kernels are compiled for; see KHG_FWD_SLOTS_LP2
#endif
#ifndef KHG_FWD_L2_WAVES
#define KHG_FWD_L2_WAVES 2  // level 2 is planned for KHG_FWD_SLOTS_L2 workgroups: two per CU
#endif
// TAP: the instantiation that also stores every coefficient as f64 (debug taps, the drop-in CompressImage).
template <int KIND, int LP, int LPC, bool TAP>
__global__ __launch_bounds__(256, KIND == KHG_FWD_PLANES2 ? KHG_FWD_L2_WAVES : LP == 2 ? KHG_FWD_LP2_WAVES : 4) void khg_fwd_kernel(KhgFwdArgs a)
{
    __shared__ KhgFwdSmemAny sh;
    khg_fwd_launch_body<KIND, LP, LPC, TAP>(a, (int)blockIdx.x, sh);
}

__global__ __launch_bounds__(256) void khg_table_kernel(KhgTableArgs a)
{
    __shared__ KhgTableSmem sh;
    khg_table_body(a, (int)blockIdx.x, sh);
}

// Start of an encode job: the dense histogram (counts 0, first raster indices "absent"), the result block and the
// state words of the single-pass packer, in one launch (three fills were three dispatches on the job's critical path).
enum { KHG_INIT_HIST_WGS = 3 * KHG_NBINS / 4 / 256 };
__global__ __launch_bounds__(256) void khg_job_init_kernel(uint32_t* hist_cnt, uint32_t* hist_first, KhgResult* res,
                                                           uint64_t* chain, uint32_t chain_pairs)
{
    const uint4 zero = {0u, 0u, 0u, 0u}, ones = {~0u, ~0u, ~0u, ~0u};
    if (blockIdx.x < KHG_INIT_HIST_WGS) {
        const uint32_t i = blockIdx.x * 256u + threadIdx.x;
        ((uint4*)hist_cnt)[i] = zero;
        ((uint4*)hist_first)[i] = ones;
        if (i < sizeof(KhgResult) / 4) ((uint32_t*)res)[i] = 0;
    } else {
        const uint32_t i = (blockIdx.x - KHG_INIT_HIST_WGS) * 256u + threadIdx.x;
        if (i < chain_pairs) ((uint4*)chain)[i] = zero;
    }
}

static void khg_launch_job_init(const KhgWorkspace& ws, int rows, int cols, hipStream_t st)
{
    KhgWorkspace none;
    memset(&none, 0, sizeof none);
    const uint32_t entries = khg_enc_args(none, rows, cols, 0, nullptr, 0).chain_entries;
    const uint32_t groups = (entries + KHG_CHAIN_GROUP - 1) / KHG_CHAIN_GROUP;
    const uint32_t pairs = (KHG_CHAIN_HEAD + 2 * (entries + groups) + 1) / 2;  // 16-byte pairs of state words
    hipLaunchKernelGGL(khg_job_init_kernel, dim3(KHG_INIT_HIST_WGS + (pairs + 255) / 256), dim3(256), 0, st, ws.hist_cnt,
                       ws.hist_first, ws.res, ws.chain, pairs);
}

// khg_packscan_body: a workgroup draws its entry when it starts.  (The tickets cost: a device-wide atomic on one
// address completes every 12 ns, so the 1024 workgroups of a round start over 12 us.  Measured and dropped: 1024
// persistent workgroups drawing entries in a loop, 95 us against 81 us -- the look-back keeps the rounds in step, so
// the draws still come all at once; drawing the next ticket ahead of time, 121 us -- an entry drawn early is
// worked on late, and its successors wait for it.)
#ifndef KHG_PS_WAVES
#define KHG_PS_WAVES 4  // waves per SIMD the register allocation aims at
#endif
__global__ __launch_bounds__(256, KHG_PS_WAVES) void khg_packscan_kernel(KhgEncArgs a)
{
    __shared__ KhgPackScanSmem sh;
    __shared__ uint32_t ticket;
    if (threadIdx.x == 0) ticket = (uint32_t)atomicAdd((unsigned long long*)a.chain, 1ull);
    __syncthreads();
    khg_packscan_body(a, ticket, sh);
}

__global__ __launch_bounds__(1024) void khg_meta_kernel(KhgEncArgs a)
{
    __shared__ KhgScanSmem sh;
    khg_meta_body(a, sh);
}

// Row-tiled images: a rank owns one chunk in `nranks`, and a workgroup that is started for a chunk only to find it
// another rank's spends its four waves on the ownership arithmetic.  So the workgroups of these two launches
// (KHG_TILE_GRID at most) first pick out their chunks -- thread t looks at chunk (j * 256 + t) * gridDim.x +
// blockIdx.x, which spreads every rank's chunks evenly over the workgroups -- and then work through that list.
enum { KHG_TILE_GRID = 1024 };
template <class BODY>
__device__ __forceinline__ void khg_for_own_chunks(const KhgEncArgs& a, uint32_t* list, uint32_t* count, BODY body)
{
    for (uint32_t first = 0; first * gridDim.x + blockIdx.x < a.nchunks; first += 256) {
        if (threadIdx.x == 0) *count = 0;
        __syncthreads();
        const uint64_t c = (uint64_t)(first + threadIdx.x) * gridDim.x + blockIdx.x;
        if (c < a.nchunks && khg_chunk_info(a, (uint32_t)c).mine) list[atomicAdd(count, 1u)] = (uint32_t)c;
        __syncthreads();
        const uint32_t n = *count;
        for (uint32_t i = 0; i < n; ++i) {
            body(list[i]);
            __syncthreads();
        }
    }
}

__global__ __launch_bounds__(256) void khg_lens_kernel(KhgEncArgs a)
{
    __shared__ KhgLensSmem sh;
    __shared__ uint32_t list[256], count;
    khg_for_own_chunks(a, list, &count, [&](uint32_t c) { khg_lens_body(a, c, sh); });
}

// Zeroes the part of a rank's share buffer the file will occupy (khg_scan_body has just fixed its length): the
// packer ORs into words that chunks share, and the ranks' shares are added up as whole words.
__global__ __launch_bounds__(256) void khg_clear_share_kernel(uint8_t* out, size_t cap, const KhgResult* res)
{
    if (!res->ok_to_pack) return;
    size_t bytes = ((size_t)res->total_len + 15) & ~(size_t)15;
    if (bytes > (cap & ~(size_t)3)) bytes = cap & ~(size_t)3;
    const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, nthreads = (size_t)gridDim.x * 256;
    if (((uintptr_t)out & 15) == 0) {
        uint4* o = (uint4*)out;
        for (size_t i = tid; i < bytes / 16; i += nthreads) o[i] = make_uint4(0, 0, 0, 0);
        for (size_t i = (bytes & ~(size_t)15) / 4 + tid; i < bytes / 4; i += nthreads) ((uint32_t*)out)[i] = 0;
    } else {
        for (size_t i = tid; i < bytes / 4; i += nthreads) ((uint32_t*)out)[i] = 0;
    }
}

__global__ __launch_bounds__(1024) void khg_scan_kernel(KhgEncArgs a)
{
    __shared__ KhgScanSmem sh;
    khg_scan_body(a, sh);
}

__global__ __launch_bounds__(256) void khg_pack_kernel(KhgEncArgs a)
{
    __shared__ KhgPackSmem sh;
    __shared__ uint32_t list[256], count;
    khg_for_own_chunks(a, list, &count, [&](uint32_t c) { khg_pack_body(a, c, sh); });
}

// rows x cols pixels of a frame, starting at image row row0.
__global__ __launch_bounds__(256) void khg_synth_kernel(uint8_t* r, uint8_t* g, uint8_t* b, int row0, int rows, int cols,
                                                        uint32_t seed, uint32_t index)
{
    size_t n = (size_t)rows * cols;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        uint32_t y = (uint32_t)(i / (size_t)cols), x = (uint32_t)(i - (size_t)y * cols);
        khg_synth_pixel((uint32_t)row0 + y, x, (uint32_t)cols, seed, index, &r[i], &g[i], &b[i]);
    }
}

__global__ __launch_bounds__(256) void khg_dec_header_kernel(KhgDecArgs a)
{
    khg_dec_header_body(a);
}

__global__ __launch_bounds__(256) void khg_dec_tree_kernel(KhgDecArgs a)
{
    __shared__ KhgDecTreeSmem sh;
    khg_dec_tree_body(a, (int)blockIdx.x, sh);
}

// MO = KHG_MAP_SMALL: planes whose longest code has at most 32 bits (three workgroups per CU);
// MO = KHG_MAXOFF: the others.  Both are launched; a workgroup of the wrong one returns at once.
template <int MO>
__global__ __launch_bounds__(KHG_MAP_THREADS) void khg_dec_map_kernel(KhgDecArgs a)
{
    __shared__ KhgDecSmem<MO> sh;
    const uint32_t blk = khg_dec_dispatch_order(a, blockIdx.x);
    if (blk != ~0u) khg_dec_map_body<MO>(a, blk, sh);
}

__global__ __launch_bounds__(256) void khg_dec_chain_kernel(KhgDecArgs a)
{
    __shared__ KhgDecChainSmem sh;
    khg_dec_chain_body(a, (int)blockIdx.x, sh);
}

// The chain of a long stream: blockIdx.x = batch of KHG_CHAIN_BATCH workgroup maps, blockIdx.y = colour component.
__global__ __launch_bounds__(256) void khg_dec_chain_maps_kernel(KhgDecArgs a)
{
    __shared__ KhgDecChainSplitSmem sh;
    khg_dec_chain_maps_body(a, (int)blockIdx.y, blockIdx.x, sh);
}
__global__ __launch_bounds__(256) void khg_dec_chain_finish_kernel(KhgDecArgs a)
{
    __shared__ KhgDecChainSplitSmem sh;
    khg_dec_chain_finish_body(a, (int)blockIdx.y, blockIdx.x, sh);
}

// Files that can hold more than KHG_CHAIN_SPLIT_FROM batches (by the bound their length gives) take the two launches
// with a workgroup per batch (an 8K frame, five batches: 8 + 10 us against 30 us); smaller ones the single launch.
#ifndef KHG_CHAIN_SPLIT_FROM
#define KHG_CHAIN_SPLIT_FROM 1
#endif
static void khg_launch_chain(hipStream_t st, const KhgDecArgs& d)
{
    const uint64_t nb = (8 * d.len + KHG_BLK_BITS - 1) / KHG_BLK_BITS + 1;
    const unsigned nbatch = (unsigned)((nb + KHG_CHAIN_BATCH - 1) / KHG_CHAIN_BATCH);
#ifdef KHG_TIMING  // the developer build keeps clock stamps in the records' spare words
    const bool split = false;
#else
    const bool split = nbatch > KHG_CHAIN_SPLIT_FROM;
#endif
    if (split) {
        hipLaunchKernelGGL(khg_dec_chain_maps_kernel, dim3(nbatch, 3), dim3(256), 0, st, d);
        hipLaunchKernelGGL(khg_dec_chain_finish_kernel, dim3(nbatch, 3), dim3(256), 0, st, d);
    } else {
        hipLaunchKernelGGL(khg_dec_chain_kernel, dim3(3), dim3(256), 0, st, d);
    }
}

__global__ __launch_bounds__(256) void khg_dec_write_kernel(KhgDecArgs a)
{
    __shared__ KhgDecWriteSmem sh;
    const uint32_t blk = khg_dec_dispatch_order(a, blockIdx.x);
    if (blk != ~0u) khg_dec_write_body(a, blk, sh);
}

// KIND: KHG_INV_PLANES2 (level 2 of the three components), KHG_INV_PLANES1 (level 1 into f64 planes),
// KHG_INV_RGB_FUSED / KHG_INV_BGR_FUSED (level 1 of the three components into the planar RGB image / a BMP pixel array).
template <int KIND>
__global__ __launch_bounds__(256) void khg_inv_kernel(KhgInvArgs a)
{
    __shared__ KhgInvSmemAny sh;
    khg_inv_launch_body<KIND>(a, (int)blockIdx.x, sh);
}

__global__ __launch_bounds__(256) void khg_index_kernel(KhgEncArgs a)
{
    __shared__ KhgIdxSmem sh;
    khg_index_body(a, blockIdx.x, sh);
}
__global__ __launch_bounds__(64) void khg_index_meta_kernel(KhgEncArgs a) { khg_index_meta_body(a); }
// Workgroups in the write launch's order, then one per plane.
__global__ __launch_bounds__(256) void khg_dec_index_kernel(KhgDecArgs a, uint32_t nwrite)
{
    uint32_t blk = blockIdx.x < nwrite ? khg_dec_dispatch_order(a, blockIdx.x) : a.nblocks + (blockIdx.x - nwrite);
    if (blk != ~0u) khg_dec_index_body(a, blk);
}

__global__ __launch_bounds__(256) void khg_vis_ycc_kernel(KhgVisYccArgs a) { khg_vis_ycc_body(a, blockIdx.x); }
__global__ __launch_bounds__(256) void khg_vis_step_kernel(KhgVisStepArgs a) { khg_vis_step_body(a, blockIdx.x); }
__global__ __launch_bounds__(256) void khg_vis_bar_kernel(KhgVisBarArgs a) { khg_vis_bar_body(a, blockIdx.x); }
__global__ __launch_bounds__(256) void khg_downsample_kernel(KhgDownArgs a) { khg_downsample_body(a, blockIdx.x); }

__global__ __launch_bounds__(256) void khg_ssim_kernel(KhgSsimArgs a)
{
    khg_ssim_body(a, blockIdx.x);
}

__global__ __launch_bounds__(KHG_OSUM_NT) void khg_ssim_mean_kernel(KhgSsimMeanArgs a)
{
    __shared__ KhgSsimMeanSmem sh;
    khg_ssim_mean_body(a, sh);
}

__global__ __launch_bounds__(256) void khg_sse_kernel(KhgSseArgs a)
{
    __shared__ KhgSseSmem sh;
    khg_sse_body(a, blockIdx.x, gridDim.x, sh);
}


// ------------------------------------------------------------------ context

#define KHG_HIP(call)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (call);                                                                         \
        if (e_ != hipSuccess) {                                                                         \
            fprintf(stderr, "khugle_hip: %s failed: %s (%s:%d)\n", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return KHG_ERR_HIP;                                                                         \
        }                                                                                               \
    } while (0)

struct khg_ctx {
    int device;
    int max_rows, max_cols;
    KhgWorkspace ws;
    KhgDecWorkspace dws;
    void* slab;          // one device allocation carved into the workspace
    size_t slab_bytes;
    KhgResult* h_res;    // pinned host copy of ws.res
    hipEvent_t done;     // recorded behind the last work enqueued on this context
    hipStream_t last_stream;
    int used;            // `done` has been recorded at least once
    int pending;         // an encode/decode was enqueued and its result not yet fetched
    int open;            // khg_begin was called and khg_end was not (yet, or ever: an entry point that failed midway)
    hipStream_t open_stream;
    // lazily allocated staging for the host-pointer entry points
    uint8_t* stage_rgb;  // 3 planes
    uint8_t* stage_comp;
    size_t stage_comp_bytes;
    double* stage_f64;   // 2 x (Y + Cb + Cr) f64 planes for the host YCbCr entry points
    double* ssim_blocks; // per-block SSIM values
    double* ssim_mean;   // [1] their mean
    unsigned long long* sse; // [3] sums of squared errors
    // optional timing of the luma forward launch inside khg_encode_dev (khg_profile_luma_forward)
    int prof_on, prof_pending, prof_launches;
    double prof_ms;      // event-to-event
    double prof_dev_ms;  // device clock, first workgroup start to last workgroup end
    hipEvent_t prof_e0, prof_e1;
    // tiled encode in progress (khg_tile_*)
    KhgTile tile;
    int tile_rows, tile_cols, tile_lvl;
    // tiled decode in progress (khg_tile_decode_map_dev .. khg_tile_decode_inverse_dev)
    const uint8_t* tdec_comp;
    size_t tdec_len;
    uint32_t tdec_nb[3];  // active workgroups of each plane's stream
    int tdec_step;        // 0 none, 1 mapped, 2 written
    int format;           // 1: the reference's file; 2: the same followed by a seek index (khg_ctx_set_format)
};

static size_t khg_align(size_t v) { return (v + 255) & ~(size_t)255; }

extern "C" const char* khg_status_string(int status)
{
    switch (status) {
    case KHG_OK: return "ok";
    case KHG_ERR_DIMS: return "bad dimensions or preset";
    case KHG_ERR_CAP: return "output buffer too small";
    case KHG_ERR_FORMAT: return "malformed .comp data";
    case KHG_ERR_RANGE: return "value out of representable range";
    case KHG_ERR_NOMEM: return "out of memory";
    case KHG_ERR_HIP: return "HIP runtime error or no device";
    case KHG_ERR_ARG: return "bad argument";
    case KHG_ERR_BUSY: return "context has a job whose result was not fetched";
    default: return "unknown status";
    }
}

extern "C" size_t khg_comp_bound(int rows, int cols)
{
    if (!khg_dims_ok(rows, cols)) return 0;
    return khg_comp_bound_impl(rows, cols);
}

extern "C" int khg_ctx_create(int device, int max_rows, int max_cols, khg_ctx** out)
{
    if (!out || !khg_dims_ok(max_rows, max_cols)) return KHG_ERR_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return KHG_ERR_HIP;
    KHG_HIP(hipSetDevice(device));
    khg_ctx* c = new (std::nothrow) khg_ctx;
    if (!c) return KHG_ERR_NOMEM;
    memset(c, 0, sizeof *c);
    c->format = 1;
    c->device = device;
    c->max_rows = max_rows;
    c->max_cols = max_cols;

    KhgWorkspaceSizes s;
    khg_workspace_sizes(max_rows, max_cols, &s);
    KhgDecWorkspaceSizes ds;
    khg_dec_workspace_sizes(max_rows, max_cols, &ds);
    size_t nblk = ((size_t)max_rows / 8) * ((size_t)max_cols / 8);
    size_t total = 0;
    auto take = [&](size_t bytes) { size_t o = total; total += khg_align(bytes); return o; };
    size_t o_ll[3], o_sym[3];
    for (int p = 0; p < 3; ++p) { o_ll[p] = take(s.ll[p]); o_sym[p] = take(s.sym[p]); }
    size_t o_hc = take(s.hist), o_hf = take(s.hist), o_tc = take(s.tab), o_ts = take(s.tab), o_tk = take(64);
    size_t o_ct = take(s.codetab), o_bs = take(s.big_scratch), o_tl = take(s.tree), o_tr = take(s.tree), o_troot = take(64);
    size_t o_cb = take(s.chunk_bits), o_co = take(s.chunk_off), o_ch = take(s.chain), o_res = take(sizeof(KhgResult));
    size_t o_lut = take(ds.lut), o_lut2 = take(ds.lut2), o_blk = take(ds.blk), o_walk = take(ds.walk), o_hdr = take(sizeof(KhgDecHeader));
    size_t o_ssim = take(nblk * sizeof(double)), o_sse = take(64), o_mean = take(64);
    c->slab_bytes = total;
    if (hipMalloc(&c->slab, total) != hipSuccess) { delete c; return KHG_ERR_NOMEM; }
    char* base = (char*)c->slab;
    for (int p = 0; p < 3; ++p) { c->ws.ll[p] = (double*)(base + o_ll[p]); c->ws.sym[p] = (int16_t*)(base + o_sym[p]); }
    c->ws.hist_cnt = (uint32_t*)(base + o_hc); c->ws.hist_first = (uint32_t*)(base + o_hf);
    c->ws.tab_cnt = (int32_t*)(base + o_tc); c->ws.tab_sym = (int32_t*)(base + o_ts); c->ws.tab_k = (int32_t*)(base + o_tk);
    c->ws.codetab = (uint64_t*)(base + o_ct); c->ws.big_scratch = (uint32_t*)(base + o_bs);
    c->ws.tree_left = (int32_t*)(base + o_tl); c->ws.tree_right = (int32_t*)(base + o_tr); c->ws.tree_root = (int32_t*)(base + o_troot);
    c->ws.chunk_bits = (uint32_t*)(base + o_cb); c->ws.chunk_off = (uint64_t*)(base + o_co);
    c->ws.chain = (uint64_t*)(base + o_ch);
    c->ws.res = (KhgResult*)(base + o_res);
    c->dws.lut = (uint32_t*)(base + o_lut);
    c->dws.lut2 = (uint32_t*)(base + o_lut2);
    c->dws.blk = (KhgBlkInfo*)(base + o_blk);
    c->dws.walk = (uint32_t*)(base + o_walk);
    c->dws.hdr = (KhgDecHeader*)(base + o_hdr);
    c->ssim_blocks = (double*)(base + o_ssim);
    c->sse = (unsigned long long*)(base + o_sse);
    c->ssim_mean = (double*)(base + o_mean);
    // hipMemset of device memory may return before the fill has run, and the fill is on the null stream, which
    // non-blocking streams do not wait for: drain it here, so that no work of the caller can overtake it.
    if (hipMemset(c->slab, 0, total) != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess ||
        hipHostMalloc((void**)&c->h_res, sizeof(KhgResult)) != hipSuccess ||
        hipEventCreateWithFlags(&c->done, hipEventDisableTiming) != hipSuccess) {
        khg_ctx_destroy(c);
        return KHG_ERR_HIP;
    }
    *out = c;
    return KHG_OK;
}

extern "C" void khg_ctx_destroy(khg_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    if (c->slab) (void)hipFree(c->slab);
    if (c->stage_rgb) (void)hipFree(c->stage_rgb);
    if (c->stage_comp) (void)hipFree(c->stage_comp);
    if (c->h_res) (void)hipHostFree(c->h_res);
    if (c->stage_f64) (void)hipFree(c->stage_f64);
    if (c->done) (void)hipEventDestroy(c->done);
    if (c->prof_e0) (void)hipEventDestroy(c->prof_e0);
    if (c->prof_e1) (void)hipEventDestroy(c->prof_e1);
    delete c;
}

static int khg_check_image(const khg_ctx* c, int rows, int cols)
{
    if (!c) return KHG_ERR_ARG;
    if (!khg_dims_ok(rows, cols) || rows > c->max_rows || cols > c->max_cols) return KHG_ERR_DIMS;
    return KHG_OK;
}

// A context owns one workspace and one result block, so its jobs must not overlap.  Every entry point that
// enqueues device work on a context calls khg_begin() before its first enqueue and khg_end() after its last:
//   - while the result of an encode / decode has not been fetched (khg_encode_result, khg_last_status) new work
//     is refused (KHG_ERR_BUSY): it would overwrite the result block and its pinned host copy;
//   - work enqueued on another stream than the context's previous work waits (on the device) for that work's
//     completion event, so a change of stream cannot let two jobs run side by side in one workspace.
//   - an entry point that fails between the two calls (a launch error, a refused argument found late) leaves the
//     job open: kernels it did launch have no completion event.  The next khg_begin sees that, waits for the stream
//     the open job was on, and starts from a clean slate (no pending result, no tiled decode in progress).
static int khg_begin(khg_ctx* c, hipStream_t st)
{
    if (c->open) {
        KHG_HIP(hipStreamSynchronize(c->open_stream));
        c->open = 0;
        c->pending = 0;
        c->tdec_step = 0;
    }
    if (c->pending) return KHG_ERR_BUSY;
    if (c->used && st != c->last_stream) KHG_HIP(hipStreamWaitEvent(st, c->done, 0));
    c->open = 1;
    c->open_stream = st;
    return KHG_OK;
}

static int khg_end(khg_ctx* c, hipStream_t st, int pending)
{
    KHG_HIP(hipEventRecord(c->done, st));
    c->last_stream = st;
    c->used = 1;
    c->pending = pending;
    c->open = 0;
    return KHG_OK;
}

// ------------------------------------------------------------------- encode

template <int KIND>
static void khg_launch_fwd(const KhgFwdArgs& a, hipStream_t st)
{
    if (a.tap) {
        if (a.lp == 2) hipLaunchKernelGGL((khg_fwd_kernel<KIND, 2, KHG_FWD_LPC, true>), dim3(a.nwg), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((khg_fwd_kernel<KIND, 1, 1, true>), dim3(a.nwg), dim3(256), 0, st, a);
    } else {
        if (a.lp == 2) hipLaunchKernelGGL((khg_fwd_kernel<KIND, 2, KHG_FWD_LPC, false>), dim3(a.nwg), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((khg_fwd_kernel<KIND, 1, 1, false>), dim3(a.nwg), dim3(256), 0, st, a);
    }
}

// Colour + both DWT levels + quantiser + histograms for the three components.
// Adds the finished luma forward launch's event-to-event time to the context's tally.
static void khg_profile_collect(khg_ctx* c)
{
    if (!c->prof_pending) return;
    c->prof_pending = 0;
    float ms = 0;
    if (hipEventElapsedTime(&ms, c->prof_e0, c->prof_e1) == hipSuccess) {
        c->prof_ms += ms;
        c->prof_dev_ms += (double)(c->h_res->prof_end - ~c->h_res->prof_not_start) / 1e5;  // 100 MHz ticks
        c->prof_launches++;
    }
}

extern "C" int khg_profile_luma_forward(khg_ctx* c, int enable)
{
    if (!c) return KHG_ERR_ARG;
    KHG_HIP(hipSetDevice(c->device));
    if (enable && !c->prof_e0) {
        KHG_HIP(hipEventCreate(&c->prof_e0));
        KHG_HIP(hipEventCreate(&c->prof_e1));
    }
    c->prof_on = enable != 0;
    c->prof_ms = 0;
    c->prof_dev_ms = 0;
    c->prof_launches = 0;
    c->prof_pending = 0;
    return KHG_OK;
}

extern "C" int khg_profile_read(khg_ctx* c, double* total_ms, double* total_device_ms, int* launches)
{
    if (!c || !total_ms || !total_device_ms || !launches) return KHG_ERR_ARG;
    *total_ms = c->prof_ms;
    *total_device_ms = c->prof_dev_ms;
    *launches = c->prof_launches;
    return KHG_OK;
}

// Same from f64 Y / Cb / Cr planes (what CompressImage receives, Compress.h:104-114).
static int khg_enqueue_forward_ycbcr(khg_ctx* c, const double* const ycbcr[3], int rows, int cols, int lvl,
                                     double* const coef[3], hipStream_t st)
{
    KhgWorkspace& ws = c->ws;
    khg_launch_job_init(ws, rows, cols, st);
    khg_launch_fwd<KHG_FWD_PLANES1>(khg_fwd_args_planes(ws, rows, cols, lvl, 1, coef, KhgTile{0, 1}, ycbcr), st);
    khg_launch_fwd<KHG_FWD_PLANES2>(khg_fwd_args_planes(ws, rows, cols, lvl, 2, coef), st);
    KHG_HIP(hipGetLastError());
    return KHG_OK;
}

static int khg_enqueue_entropy(khg_ctx* c, int rows, int cols, int lvl, uint8_t* d_out, size_t cap, hipStream_t st)
{
    KhgTableArgs t = khg_table_args(c->ws);
    hipLaunchKernelGGL(khg_table_kernel, dim3(3), dim3(256), 0, st, t);
    KhgEncArgs e = khg_enc_args(c->ws, rows, cols, lvl, d_out, cap);
    hipLaunchKernelGGL(khg_packscan_kernel, dim3(e.chain_entries), dim3(256), 0, st, e);
    hipLaunchKernelGGL(khg_meta_kernel, dim3(1), dim3(1024), 0, st, e);
    if (c->format == 2) {  // the seek index behind the file (khg_index.h)
        hipLaunchKernelGGL(khg_index_kernel, dim3(e.chain_entries), dim3(256), 0, st, e);
        hipLaunchKernelGGL(khg_index_meta_kernel, dim3(1), dim3(64), 0, st, e);
    }
    KHG_HIP(hipGetLastError());
    KHG_HIP(hipMemcpyAsync(c->h_res, c->ws.res, sizeof(KhgResult), hipMemcpyDeviceToHost, st));
    return khg_end(c, st, 1);
}

static int khg_enqueue_forward(khg_ctx* c, const uint8_t* r, const uint8_t* g, const uint8_t* b, int rows, int cols,
                               int lvl, double* const coef[3], hipStream_t st, KhgTile tile = KhgTile{0, 1},
                               int bgr_stride = 0)
{
    KhgWorkspace& ws = c->ws;
    khg_launch_job_init(ws, rows, cols, st);
    KhgFwdArgs a = khg_fwd_args_rgb(ws, r, g, b, rows, cols, lvl, coef, tile, false, bgr_stride);
    if (c->prof_on) {
        KHG_HIP(hipEventRecord(c->prof_e0, st));
        a.prof = ws.res;
    }
    if (bgr_stride) khg_launch_fwd<KHG_FWD_BGR>(a, st);
    else khg_launch_fwd<KHG_FWD_RGB>(a, st);
    if (c->prof_on) {
        KHG_HIP(hipEventRecord(c->prof_e1, st));
        c->prof_pending = 1;
    }
    khg_launch_fwd<KHG_FWD_PLANES2>(khg_fwd_args_planes(ws, rows, cols, lvl, 2, coef, tile), st);
    KHG_HIP(hipGetLastError());
    return KHG_OK;
}

extern "C" int khg_encode_dev(khg_ctx* c, const uint8_t* d_r, const uint8_t* d_g, const uint8_t* d_b, int rows,
                              int cols, int lvl, uint8_t* d_out, size_t cap, void* stream)
{
    int rc = khg_check_image(c, rows, cols);
    if (rc) return rc;
    if (lvl < 0 || lvl > 2) return KHG_ERR_DIMS;
    if (!d_r || !d_g || !d_b || !d_out || ((uintptr_t)d_out & 3)) return KHG_ERR_ARG;
    if (cap < 40) return KHG_ERR_CAP;
    hipStream_t st = (hipStream_t)stream;
    KHG_HIP(hipSetDevice(c->device));
    if ((rc = khg_begin(c, st))) return rc;
    rc = khg_enqueue_forward(c, d_r, d_g, d_b, rows, cols, lvl, nullptr, st);
    if (rc) return rc;
    return khg_enqueue_entropy(c, rows, cols, lvl, d_out, cap, st);
}

extern "C" int khg_encode_bgr_dev(khg_ctx* c, const uint8_t* d_pixels, size_t stride, int rows, int cols, int lvl,
                                  uint8_t* d_out, size_t cap, void* stream)
{
    int rc = khg_check_image(c, rows, cols);
    if (rc) return rc;
    if (lvl < 0 || lvl > 2) return KHG_ERR_DIMS;
    if (!d_pixels || !d_out || ((uintptr_t)d_out & 3)) return KHG_ERR_ARG;
    if (stride < 3 * (size_t)cols || stride * (size_t)rows >= ((size_t)1 << 32)) return KHG_ERR_DIMS;  // 32-bit byte offsets
    if (cap < 40) return KHG_ERR_CAP;
    hipStream_t st = (hipStream_t)stream;
    KHG_HIP(hipSetDevice(c->device));
    if ((rc = khg_begin(c, st))) return rc;
    rc = khg_enqueue_forward(c, d_pixels, nullptr, nullptr, rows, cols, lvl, nullptr, st, KhgTile{0, 1}, (int)stride);
    if (rc) return rc;
    return khg_enqueue_entropy(c, rows, cols, lvl, d_out, cap, st);
}

extern "C" int khg_encode_ycbcr_dev(khg_ctx* c, const double* d_y, const double* d_cb, const double* d_cr, int rows,
                                    int cols, int lvl, uint8_t* d_out, size_t cap, double* d_coef_y,
                                    double* d_coef_cb, double* d_coef_cr, void* stream)
{
    int rc = khg_check_image(c, rows, cols);
    if (rc) return rc;
    if (lvl < 0 || lvl > 2) return KHG_ERR_DIMS;
    if (!d_y || !d_cb || !d_cr || !d_out || ((uintptr_t)d_out & 3)) return KHG_ERR_ARG;
    if (cap < 40) return KHG_ERR_CAP;
    hipStream_t st = (hipStream_t)stream;
    KHG_HIP(hipSetDevice(c->device));
    const double* src[3] = {d_y, d_cb, d_cr};
    double* coef[3] = {d_coef_y, d_coef_cb, d_coef_cr};
    const bool tap = d_coef_y && d_coef_cb && d_coef_cr;
    if ((rc = khg_begin(c, st))) return rc;
    rc = khg_enqueue_forward_ycbcr(c, src, rows, cols, lvl, tap ? coef : nullptr, st);
    if (rc) return rc;
    return khg_enqueue_entropy(c, rows, cols, lvl, d_out, cap, st);
}

extern "C" int khg_ctx_set_format(khg_ctx* c, int format)
{
    if (!c || (format != 1 && format != 2)) return KHG_ERR_ARG;
    c->format = format;
    return KHG_OK;
}

extern "C" size_t khg_index_bound(int rows, int cols)
{
    if (!khg_dims_ok(rows, cols)) return 0;
    // three streams of at most khg_comp_bound bytes together, each with one partly used entry at its end
    const uint64_t entries = 8ull * khg_comp_bound_impl(rows, cols) / KHG_SUB_BITS + 3;
    return (size_t)(8 + khg_idx_bytes(entries));
}

extern "C" int khg_encode_result(khg_ctx* c, size_t* out_len)
{
    if (!c) return KHG_ERR_ARG;
    if (c->pending != 1) return KHG_ERR_ARG;
    KHG_HIP(hipEventSynchronize(c->done));
    c->pending = 0;
    khg_profile_collect(c);
    if (out_len) *out_len = (size_t)c->h_res->total_len;
    return c->h_res->status;
}

static int khg_stage(khg_ctx* c, size_t comp_bytes)
{
    if (!c->stage_rgb) {
        size_t n = (size_t)c->max_rows * c->max_cols;
        if (hipMalloc((void**)&c->stage_rgb, 3 * n) != hipSuccess) return KHG_ERR_NOMEM;
    }
    if (c->stage_comp_bytes < comp_bytes) {
        if (c->stage_comp) (void)hipFree(c->stage_comp);
        c->stage_comp = nullptr;
        c->stage_comp_bytes = 0;
        if (hipMalloc((void**)&c->stage_comp, comp_bytes) != hipSuccess) return KHG_ERR_NOMEM;
        c->stage_comp_bytes = comp_bytes;
    }
    return KHG_OK;
}

extern "C" int khg_encode(khg_ctx* c, const uint8_t* r, const uint8_t* g, const uint8_t* b, int rows, int cols, int lvl,
                          uint8_t* out, size_t cap, size_t* out_len)
{
    int rc = khg_check_image(c, rows, cols);
    if (rc) return rc;
    if (!r || !g || !b || !out) return KHG_ERR_ARG;
    KHG_HIP(hipSetDevice(c->device));
    size_t n = (size_t)rows * cols;
    size_t dcap = khg_comp_bound_impl(rows, cols);
    if ((rc = khg_stage(c, dcap))) return rc;
    uint8_t* d = c->stage_rgb;
    KHG_HIP(hipMemcpy(d, r, n, hipMemcpyHostToDevice));
    KHG_HIP(hipMemcpy(d + n, g, n, hipMemcpyHostToDevice));
    KHG_HIP(hipMemcpy(d + 2 * n, b, n, hipMemcpyHostToDevice));
    rc = khg_encode_dev(c, d, d + n, d + 2 * n, rows, cols, lvl, c->stage_comp, dcap, nullptr);
    if (rc) return rc;
    size_t len = 0;
    rc = khg_encode_result(c, &len);
    if (out_len) *out_len = len;
    if (rc) return rc;
    if (len > cap) return KHG_ERR_CAP;
    KHG_HIP(hipMemcpy(out, c->stage_comp, len, hipMemcpyDeviceToHost));
    return KHG_OK;
}

// -------------------------------------------------------------- parity taps

extern "C" int khg_forward_planes_dev(khg_ctx* c, const uint8_t* d_r, const uint8_t* d_g, const uint8_t* d_b, int rows,
                                      int cols, double* d_y, double* d_cb, double* d_cr, void* stream)
{
    int rc = khg_check_image(c, rows, cols);
    if (rc) return rc;
    if (!d_y || !d_cb || !d_cr) return KHG_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    KHG_HIP(hipSetDevice(c->device));
    double* coef[3] = {d_y, d_cb, d_cr};
    if ((rc = khg_begin(c, st))) return rc;
    rc = khg_enqueue_forward(c, d_r, d_g, d_b, rows, cols, 0, coef, st);
    if (rc) return rc;
    if ((rc = khg_end(c, st, 0))) return rc;
    KHG_HIP(hipStreamSynchronize(st));
    return KHG_OK;
}

extern "C" int khg_symbols_dev(khg_ctx* c, const uint8_t* d_r, const uint8_t* d_g, const uint8_t* d_b, int rows,
                               int cols, int lvl, int16_t* d_qy, int16_t* d_qcb, int16_t* d_qcr, void* stream)
{
    int rc = khg_check_image(c, rows, cols);
    if (rc) return rc;
    if (lvl < 0 || lvl > 2) return KHG_ERR_DIMS;
    hipStream_t st = (hipStream_t)stream;
    KHG_HIP(hipSetDevice(c->device));
    if ((rc = khg_begin(c, st))) return rc;
    rc = khg_enqueue_forward(c, d_r, d_g, d_b, rows, cols, lvl, nullptr, st);
    if (rc) return rc;
    size_t ny = (size_t)rows * cols;
    int16_t* dst[3] = {d_qy, d_qcb, d_qcr};
    for (int p = 0; p < 3; ++p)
        if (dst[p])
            KHG_HIP(hipMemcpyAsync(dst[p], c->ws.sym[p], (p ? ny / 4 : ny) * sizeof(int16_t), hipMemcpyDeviceToDevice, st));
    KHG_HIP(hipMemcpyAsync(c->h_res, c->ws.res, sizeof(KhgResult), hipMemcpyDeviceToHost, st));
    if ((rc = khg_end(c, st, 0))) return rc;
    KHG_HIP(hipStreamSynchronize(st));
    return c->h_res->status;
}

extern "C" int khg_time_luma_forward_dev(khg_ctx* c, const uint8_t* d_r, const uint8_t* d_g, const uint8_t* d_b,
                                         int rows, int cols, int lvl, int reps, float* avg_ms, void* stream)
{
    int rc = khg_check_image(c, rows, cols);
    if (rc) return rc;
    if (lvl < 0 || lvl > 2 || reps < 1 || !avg_ms) return KHG_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    KHG_HIP(hipSetDevice(c->device));
    if ((rc = khg_begin(c, st))) return rc;
    KhgWorkspace& ws = c->ws;
    khg_launch_job_init(ws, rows, cols, st);
    KhgFwdArgs a = khg_fwd_args_rgb(ws, d_r, d_g, d_b, rows, cols, lvl, nullptr);
    hipEvent_t e0, e1;
    KHG_HIP(hipEventCreate(&e0));
    KHG_HIP(hipEventCreate(&e1));
    khg_launch_fwd<KHG_FWD_RGB>(a, st);  // warm-up
    KHG_HIP(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i) khg_launch_fwd<KHG_FWD_RGB>(a, st);
    KHG_HIP(hipEventRecord(e1, st));
    if ((rc = khg_end(c, st, 0))) return rc;
    KHG_HIP(hipEventSynchronize(e1));
    float ms = 0;
    KHG_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *avg_ms = ms / reps;
    return KHG_OK;
}

extern "C" int khg_debug_poison(khg_ctx* c, int byte)
{
    if (!c) return KHG_ERR_ARG;
    if (c->pending) return KHG_ERR_BUSY;
    KHG_HIP(hipSetDevice(c->device));
    KHG_HIP(hipDeviceSynchronize());  // whatever stream the context's last work ran on
    KHG_HIP(hipMemset(c->slab, byte, c->slab_bytes));
    KHG_HIP(hipDeviceSynchronize());
    return KHG_OK;
}

extern "C" int khg_synth_rows_dev(uint8_t* d_r, uint8_t* d_g, uint8_t* d_b, int row0, int rows, int cols,
                                  uint32_t seed, uint32_t index, void* stream);

extern "C" int khg_synth_dev(uint8_t* d_r, uint8_t* d_g, uint8_t* d_b, int rows, int cols, uint32_t seed,
                             uint32_t index, void* stream)
{
    if (!d_r || !d_g || !d_b || rows <= 0 || cols <= 0) return KHG_ERR_ARG;
    return khg_synth_rows_dev(d_r, d_g, d_b, 0, rows, cols, seed, index, stream);
}

extern "C" int khg_synth_rows_dev(uint8_t* d_r, uint8_t* d_g, uint8_t* d_b, int row0, int rows, int cols,
                                  uint32_t seed, uint32_t index, void* stream)
{
    if (!d_r || !d_g || !d_b || row0 < 0 || rows <= 0 || cols <= 0) return KHG_ERR_ARG;
    size_t n = (size_t)rows * cols;
    unsigned grid = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(khg_synth_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, d_r, d_g, d_b, row0, rows, cols,
                       seed, index);
    KHG_HIP(hipGetLastError());
    return KHG_OK;
}

#include "khg_host_decode.inl"
#include "khg_host_tile.inl"
#include "khg_host_comm.inl"
#include "khg_host_batch.inl"
#include "khg_host_visual.inl"
