// khg_host_decode.inl -- host side of the decode path and the quality metrics
// (included by khugle_hip.hip).

extern "C" int khg_peek(const uint8_t* comp, size_t len, khg_header* hdr)
{
    if (!comp || !hdr) return KHG_ERR_ARG;
    if (len < 36) return KHG_ERR_FORMAT;
    auto rd = [&](size_t o) {
        return (int32_t)((uint32_t)comp[o] | ((uint32_t)comp[o + 1] << 8) | ((uint32_t)comp[o + 2] << 16) |
                         ((uint32_t)comp[o + 3] << 24));
    };
    // Reader.h:5-23
    hdr->y_len = rd(0); hdr->cb_len = rd(4); hdr->cr_len = rd(8);
    hdr->rle_y_len = rd(12); hdr->rle_cb_len = rd(16); hdr->rle_cr_len = rd(20);
    hdr->h = rd(24); hdr->w = rd(28);
    hdr->y_pad = comp[32]; hdr->cb_pad = comp[33]; hdr->cr_pad = comp[34]; hdr->lvl = comp[35];
    if (!khg_dims_ok(hdr->h, hdr->w) || hdr->lvl > 2) return KHG_ERR_FORMAT;
    const int32_t lens[3] = {hdr->y_len, hdr->cb_len, hdr->cr_len};
    const int32_t ks[3] = {hdr->rle_y_len, hdr->rle_cb_len, hdr->rle_cr_len};
    const uint8_t pads[3] = {hdr->y_pad, hdr->cb_pad, hdr->cr_pad};
    uint64_t need = 36;
    for (int p = 0; p < 3; ++p) {
        if (lens[p] < 0 || (lens[p] & 7) || pads[p] > 7 || pads[p] > lens[p]) return KHG_ERR_FORMAT;
        if (ks[p] < 1) return KHG_ERR_FORMAT;
        need += (uint64_t)(lens[p] >> 3) + 8ull * (uint64_t)ks[p];
    }
    if (need > len) return KHG_ERR_FORMAT;
    return KHG_OK;
}

// The inverse transform launches of a decode (symbols in the workspace): Decompress.h:274-280, Colorspace.h:25-36.
// bgr_stride != 0: d_r is a BMP pixel array with that many bytes per row (whole images only), d_g and d_b are unused.
static void khg_enqueue_inverse(khg_ctx* c, int rows, int cols, uint8_t* d_r, uint8_t* d_g, uint8_t* d_b, hipStream_t st,
                                KhgTile tile, double* const ycbcr[3], size_t bgr_stride = 0)
{
    {  // Decompress.h:274-276: level 2 of the three components, one launch
        const KhgInvArgs a = khg_inv_args(c->ws, c->dws, rows, cols, 0, 3, 2, tile);
        hipLaunchKernelGGL((khg_inv_kernel<KHG_INV_PLANES2>), dim3(a.nwg), dim3(256), 0, st, a);
    }
    if (ycbcr) {  // Decompress.h:278-280 into the caller's f64 planes
        const KhgInvArgs a = khg_inv_args(c->ws, c->dws, rows, cols, 0, 3, 1, tile, ycbcr);
        hipLaunchKernelGGL((khg_inv_kernel<KHG_INV_PLANES1>), dim3(a.nwg), dim3(256), 0, st, a);
    } else {  // level 1 of Y, Cb and Cr with Colorspace.h:25-36 + Main.cpp:307-315, one launch
        KhgInvArgs ay = khg_inv_args_fused(c->ws, c->dws, rows, cols, tile);
        ay.r = d_r; ay.g = d_g; ay.b = d_b;
        ay.bgr_stride = bgr_stride;
        if (bgr_stride) hipLaunchKernelGGL((khg_inv_kernel<KHG_INV_BGR_FUSED>), dim3(ay.nwg), dim3(256), 0, st, ay);
        else hipLaunchKernelGGL((khg_inv_kernel<KHG_INV_RGB_FUSED>), dim3(ay.nwg), dim3(256), 0, st, ay);
    }
}

// ycbcr != null: stop before the colour conversion and leave the f64 planes there (DecompressImage's outputs).
static int khg_decode_enqueue(khg_ctx* c, const uint8_t* d_comp, size_t len, int rows, int cols, uint8_t* d_r,
                              uint8_t* d_g, uint8_t* d_b, void* stream, KhgTile tile, double* const ycbcr[3] = nullptr,
                              size_t bgr_stride = 0, bool coeff_picture = false, bool indexed = false)
{
    int rc = khg_check_image(c, rows, cols);
    if (rc) return rc;
    if (!d_comp || ((uintptr_t)d_comp & 3)) return KHG_ERR_ARG;
    if (bgr_stride) {  // 16-bit stores at even byte offsets of the array
        if (!d_r || ((uintptr_t)d_r & 1) || (bgr_stride & 1)) return KHG_ERR_ARG;
    } else if (!ycbcr && (!d_r || !d_g || !d_b || ((uintptr_t)d_r & 7) || ((uintptr_t)d_g & 7) || ((uintptr_t)d_b & 7)))
        return KHG_ERR_ARG;
    if (len < 36 + 24) return KHG_ERR_FORMAT;
    hipStream_t st = (hipStream_t)stream;
    KHG_HIP(hipSetDevice(c->device));
    if ((rc = khg_begin(c, st))) return rc;
    const KhgDecArgs d = khg_dec_args(c->ws, c->dws, d_comp, len, rows, cols);
    hipLaunchKernelGGL(khg_dec_header_kernel, dim3(1), dim3(256), 0, st, d);
    hipLaunchKernelGGL(khg_dec_tree_kernel, dim3(3), dim3(256), 0, st, d);
    // the three streams together are shorter than the file, and each ends inside one more workgroup at most
    uint64_t spanned = (8ull * len) / KHG_BLK_BITS + 3;
    const unsigned grid_dec = (unsigned)(spanned < d.nblocks ? spanned : d.nblocks);
    if (indexed) {  // the file says where codewords start (khg_index.h); KHG_ERR_FORMAT if it does not
        hipLaunchKernelGGL(khg_dec_index_kernel, dim3(grid_dec + 3), dim3(256), 0, st, d, grid_dec);
    } else {
        hipLaunchKernelGGL((khg_dec_map_kernel<KHG_MAP_SMALL>), dim3(grid_dec), dim3(KHG_MAP_THREADS), 0, st, d);
        hipLaunchKernelGGL((khg_dec_map_kernel<KHG_MAXOFF>), dim3(grid_dec), dim3(KHG_MAP_THREADS), 0, st, d);
        khg_launch_chain(st, d);
    }
    hipLaunchKernelGGL(khg_dec_write_kernel, dim3(grid_dec), dim3(256), 0, st, d);
    if (coeff_picture) {  // Decompress.h:234-262: the dequantised coefficients through the colour conversion, no IWT2D
        KhgVisYccArgs v;
        memset(&v, 0, sizeof v);
        v.rows = rows; v.cols = cols;
        for (int p = 0; p < 3; ++p) v.sym[p] = c->ws.sym[p];
        v.hdr = c->dws.hdr;
        for (int lvl = 0; lvl < 3; ++lvl) {
            khg_make_quant(rows, cols, lvl, 0, &v.q[0][lvl]);
            khg_make_quant(rows, cols, lvl, 1, &v.q[1][lvl]);
        }
        v.r = d_r; v.g = d_g; v.b = d_b;
        hipLaunchKernelGGL(khg_vis_ycc_kernel, dim3((unsigned)(((size_t)rows * cols + 255) / 256)), dim3(256), 0, st, v);
    } else {
        khg_enqueue_inverse(c, rows, cols, d_r, d_g, d_b, st, tile, ycbcr, bgr_stride);
    }
    KHG_HIP(hipGetLastError());
    KHG_HIP(hipMemcpyAsync(c->h_res, c->ws.res, sizeof(KhgResult), hipMemcpyDeviceToHost, st));
    return khg_end(c, st, 2);
}

extern "C" int khg_decode_dev(khg_ctx* c, const uint8_t* d_comp, size_t len, int rows, int cols, uint8_t* d_r,
                              uint8_t* d_g, uint8_t* d_b, void* stream)
{
    return khg_decode_enqueue(c, d_comp, len, rows, cols, d_r, d_g, d_b, stream, KhgTile{0, 1});
}

extern "C" int khg_decode_bgr_dev(khg_ctx* c, const uint8_t* d_comp, size_t len, int rows, int cols, uint8_t* d_pixels,
                                  size_t stride, void* stream)
{
    if (!d_pixels) return KHG_ERR_ARG;
    if (stride < 3 * (size_t)(cols > 0 ? cols : 0)) return KHG_ERR_DIMS;
    return khg_decode_enqueue(c, d_comp, len, rows, cols, d_pixels, nullptr, nullptr, stream, KhgTile{0, 1}, nullptr, stride);
}

extern "C" int khg_decode_indexed_dev(khg_ctx* c, const uint8_t* d_comp, size_t len, int rows, int cols, uint8_t* d_r,
                                      uint8_t* d_g, uint8_t* d_b, void* stream)
{
    return khg_decode_enqueue(c, d_comp, len, rows, cols, d_r, d_g, d_b, stream, KhgTile{0, 1}, nullptr, 0, false, true);
}

extern "C" int khg_decode_ycbcr_dev(khg_ctx* c, const uint8_t* d_comp, size_t len, int rows, int cols, double* d_y,
                                    double* d_cb, double* d_cr, void* stream)
{
    if (!d_y || !d_cb || !d_cr) return KHG_ERR_ARG;
    double* planes[3] = {d_y, d_cb, d_cr};
    return khg_decode_enqueue(c, d_comp, len, rows, cols, nullptr, nullptr, nullptr, stream, KhgTile{0, 1}, planes);
}

extern "C" int khg_last_status(khg_ctx* c)
{
    if (!c) return KHG_ERR_ARG;
    if (!c->pending) return KHG_OK;
    KHG_HIP(hipEventSynchronize(c->done));
    c->pending = 0;
    return c->h_res->status;
}

extern "C" int khg_decode(khg_ctx* c, const uint8_t* comp, size_t len, uint8_t* r, uint8_t* g, uint8_t* b)
{
    if (!c || !comp || !r || !g || !b) return KHG_ERR_ARG;
    khg_header hd;
    int rc = khg_peek(comp, len, &hd);
    if (rc) return rc;
    rc = khg_check_image(c, hd.h, hd.w);
    if (rc) return rc;
    KHG_HIP(hipSetDevice(c->device));
    size_t n = (size_t)hd.h * hd.w;
    size_t padded = (len + 3) & ~(size_t)3;
    if ((rc = khg_stage(c, padded))) return rc;
    KHG_HIP(hipMemcpy(c->stage_comp, comp, len, hipMemcpyHostToDevice));
    uint8_t* d = c->stage_rgb;
    rc = khg_decode_dev(c, c->stage_comp, len, hd.h, hd.w, d, d + n, d + 2 * n, nullptr);
    if (rc) return rc;
    rc = khg_last_status(c);
    if (rc) return rc;
    KHG_HIP(hipMemcpy(r, d, n, hipMemcpyDeviceToHost));
    KHG_HIP(hipMemcpy(g, d + n, n, hipMemcpyDeviceToHost));
    KHG_HIP(hipMemcpy(b, d + 2 * n, n, hipMemcpyDeviceToHost));
    return KHG_OK;
}

// ---- host-pointer forms of the f64 plane entry points: CompressImage (Compress.h:104) and DecompressImage
// (Decompress.h:7) as the reference calls them, on contiguous planes.
static int khg_stage_f64(khg_ctx* c)
{
    if (c->stage_f64) return KHG_OK;
    size_t n = (size_t)c->max_rows * c->max_cols;
    if (hipMalloc((void**)&c->stage_f64, 2 * (n + n / 2) * sizeof(double)) != hipSuccess) return KHG_ERR_NOMEM;
    return KHG_OK;
}

extern "C" int khg_encode_ycbcr(khg_ctx* c, double* y, double* cb, double* cr, int rows, int cols, int lvl,
                                uint8_t* out, size_t cap, size_t* out_len, int write_back)
{
    int rc = khg_check_image(c, rows, cols);
    if (rc) return rc;
    if (!y || !cb || !cr || !out) return KHG_ERR_ARG;
    KHG_HIP(hipSetDevice(c->device));
    const size_t n = (size_t)rows * cols, nc = n / 4;
    size_t dcap = khg_comp_bound_impl(rows, cols);
    if ((rc = khg_stage(c, dcap)) || (rc = khg_stage_f64(c))) return rc;
    double* in[3] = {c->stage_f64, c->stage_f64 + n, c->stage_f64 + n + nc};
    double* co[3] = {in[2] + nc, in[2] + nc + n, in[2] + nc + n + nc};
    KHG_HIP(hipMemcpy(in[0], y, n * 8, hipMemcpyHostToDevice));
    KHG_HIP(hipMemcpy(in[1], cb, nc * 8, hipMemcpyHostToDevice));
    KHG_HIP(hipMemcpy(in[2], cr, nc * 8, hipMemcpyHostToDevice));
    rc = khg_encode_ycbcr_dev(c, in[0], in[1], in[2], rows, cols, lvl, c->stage_comp, dcap, write_back ? co[0] : nullptr,
                              write_back ? co[1] : nullptr, write_back ? co[2] : nullptr, nullptr);
    if (rc) return rc;
    size_t len = 0;
    rc = khg_encode_result(c, &len);
    if (rc) return rc;
    if (len > cap) return KHG_ERR_CAP;
    KHG_HIP(hipMemcpy(out, c->stage_comp, len, hipMemcpyDeviceToHost));
    if (write_back) {  // CompressImage leaves the DWT coefficients in its input planes (Compress.h:108-114)
        KHG_HIP(hipMemcpy(y, co[0], n * 8, hipMemcpyDeviceToHost));
        KHG_HIP(hipMemcpy(cb, co[1], nc * 8, hipMemcpyDeviceToHost));
        KHG_HIP(hipMemcpy(cr, co[2], nc * 8, hipMemcpyDeviceToHost));
    }
    if (out_len) *out_len = len;
    return KHG_OK;
}

extern "C" int khg_decode_ycbcr(khg_ctx* c, const uint8_t* comp, size_t len, double* y, double* cb, double* cr)
{
    if (!c || !comp || !y || !cb || !cr) return KHG_ERR_ARG;
    khg_header hd;
    int rc = khg_peek(comp, len, &hd);
    if (rc) return rc;
    rc = khg_check_image(c, hd.h, hd.w);
    if (rc) return rc;
    KHG_HIP(hipSetDevice(c->device));
    const size_t n = (size_t)hd.h * hd.w, nc = n / 4;
    size_t padded = (len + 3) & ~(size_t)3;
    if ((rc = khg_stage(c, padded)) || (rc = khg_stage_f64(c))) return rc;
    KHG_HIP(hipMemcpy(c->stage_comp, comp, len, hipMemcpyHostToDevice));
    double* d[3] = {c->stage_f64, c->stage_f64 + n, c->stage_f64 + n + nc};
    rc = khg_decode_ycbcr_dev(c, c->stage_comp, len, hd.h, hd.w, d[0], d[1], d[2], nullptr);
    if (rc) return rc;
    rc = khg_last_status(c);
    if (rc) return rc;
    KHG_HIP(hipMemcpy(y, d[0], n * 8, hipMemcpyDeviceToHost));
    KHG_HIP(hipMemcpy(cb, d[1], nc * 8, hipMemcpyDeviceToHost));
    KHG_HIP(hipMemcpy(cr, d[2], nc * 8, hipMemcpyDeviceToHost));
    return KHG_OK;
}

// ------------------------------------------------------------------ metrics

// Per-block values into `blocks`, then their raster-order mean into d_mean[0] (both device memory).
static int khg_ssim_enqueue(const uint8_t* d_a, const uint8_t* d_b, int width, int height, double* blocks, double* d_mean,
                            hipStream_t st)
{
    const size_t nblk = (size_t)(width / 8) * (height / 8);
    KhgSsimArgs a;
    a.a = d_a; a.b = d_b; a.width = width; a.height = height; a.blocks = blocks;
    hipLaunchKernelGGL(khg_ssim_kernel, dim3((unsigned)((nblk + 255) / 256)), dim3(256), 0, st, a);
    KhgSsimMeanArgs m;
    m.blocks = blocks; m.nblk = nblk; m.out = d_mean; m.stat = nullptr;
    hipLaunchKernelGGL(khg_ssim_mean_kernel, dim3(1), dim3(KHG_OSUM_NT), 0, st, m);
    KHG_HIP(hipGetLastError());
    return KHG_OK;
}

extern "C" int khg_ssim_dev(khg_ctx* c, const uint8_t* d_a, const uint8_t* d_b, int width, int height, double* ssim,
                            void* stream)
{
    if (!c || !d_a || !d_b || !ssim) return KHG_ERR_ARG;
    if (width < 8 || height < 8 || width > c->max_cols || height > c->max_rows) return KHG_ERR_DIMS;
    hipStream_t st = (hipStream_t)stream;
    KHG_HIP(hipSetDevice(c->device));
    int rc = khg_ssim_enqueue(d_a, d_b, width, height, c->ssim_blocks, c->ssim_mean, st);
    if (rc) return rc;
    KHG_HIP(hipMemcpyAsync(ssim, c->ssim_mean, sizeof(double), hipMemcpyDeviceToHost, st));
    KHG_HIP(hipStreamSynchronize(st));
    return KHG_OK;
}

extern "C" int khg_ordered_mean_dev(const double* d_values, size_t n, double* d_mean, void* stream)
{
    if (!d_values || !d_mean || n == 0) return KHG_ERR_ARG;
    KhgSsimMeanArgs m;
    m.blocks = d_values; m.nblk = n; m.out = d_mean; m.stat = nullptr;
    hipLaunchKernelGGL(khg_ssim_mean_kernel, dim3(1), dim3(KHG_OSUM_NT), 0, (hipStream_t)stream, m);
    KHG_HIP(hipGetLastError());
    return KHG_OK;
}

// ---- the entry point that allocates nothing: per-block values go to a fixed array of the code object
enum { KHG_SSIM_STATIC_BLOCKS = (7680 / 8) * (4320 / 8), KHG_MAX_DEVICES = 64 };
static __device__ __align__(256) double khg_ssim_scratch[KHG_SSIM_STATIC_BLOCKS];
static double* g_ssim_scratch_ptr[KHG_MAX_DEVICES];  // device address of the array on each device, once looked up

// The array's address on the current device (each device holds its own copy at its own address), looked up once.
static int khg_ssim_scratch_of_current_device(int* dev_out, double** scratch)
{
    int dev = -1;
    KHG_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= KHG_MAX_DEVICES) return KHG_ERR_HIP;
    double* sc = __atomic_load_n(&g_ssim_scratch_ptr[dev], __ATOMIC_ACQUIRE);
    if (!sc) {
        void* p = nullptr;
        KHG_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(khg_ssim_scratch)));
        if (!p) return KHG_ERR_HIP;
        sc = (double*)p;
        __atomic_store_n(&g_ssim_scratch_ptr[dev], sc, __ATOMIC_RELEASE);
    }
    *dev_out = dev;
    *scratch = sc;
    return KHG_OK;
}

extern "C" int khg_ssim_static_prepare(void)
{
    int dev;
    double* scratch;
    return khg_ssim_scratch_of_current_device(&dev, &scratch);
}

extern "C" int khg_ssim_static_dev(const uint8_t* d_a, const uint8_t* d_b, int width, int height, double* d_ssim,
                                   void* stream)
{
    if (!d_a || !d_b || !d_ssim) return KHG_ERR_ARG;
    if (width < 8 || height < 8) return KHG_ERR_DIMS;
    if ((size_t)(width / 8) * (size_t)(height / 8) > (size_t)KHG_SSIM_STATIC_BLOCKS) return KHG_ERR_DIMS;  // beyond the fixed array
    int dev = -1;
    double* scratch = nullptr;
    int rc = khg_ssim_scratch_of_current_device(&dev, &scratch);
    if (rc) return rc;
    if (stream) {  // the array used is the current device's: a stream of another device would write through a foreign address
        int sdev = -1;
        KHG_HIP(hipStreamGetDevice((hipStream_t)stream, &sdev));
        if (sdev != dev) return KHG_ERR_ARG;
    }
    return khg_ssim_enqueue(d_a, d_b, width, height, scratch, d_ssim, (hipStream_t)stream);
}

extern "C" int khg_psnr_dev(khg_ctx* c, const uint8_t* d_ir, const uint8_t* d_ig, const uint8_t* d_ib,
                            const uint8_t* d_or, const uint8_t* d_og, const uint8_t* d_ob, int width, int height,
                            double* psnr, void* stream)
{
    if (!c || !d_ir || !d_ig || !d_ib || !d_or || !d_og || !d_ob || !psnr) return KHG_ERR_ARG;
    if (width <= 0 || height <= 0) return KHG_ERR_DIMS;
    hipStream_t st = (hipStream_t)stream;
    KHG_HIP(hipSetDevice(c->device));
    KhgSseArgs a;
    a.in[0] = d_ir; a.in[1] = d_ig; a.in[2] = d_ib;
    a.out[0] = d_or; a.out[1] = d_og; a.out[2] = d_ob;
    a.n = (size_t)width * height;
    a.sse = c->sse;
    KHG_HIP(hipMemsetAsync(c->sse, 0, 3 * sizeof(unsigned long long), st));
    unsigned grid = (unsigned)((a.n + 255) / 256 < 2048 ? (a.n + 255) / 256 : 2048);
    hipLaunchKernelGGL(khg_sse_kernel, dim3(grid), dim3(256), 0, st, a);
    KHG_HIP(hipGetLastError());
    unsigned long long h[3];
    KHG_HIP(hipMemcpyAsync(h, c->sse, sizeof h, hipMemcpyDeviceToHost, st));
    KHG_HIP(hipStreamSynchronize(st));
    // KhuGleBase.cpp:609-626
    double MseR = (double)h[0], MseG = (double)h[1], MseB = (double)h[2];
    MseR /= width * height;
    MseG /= width * height;
    MseB /= width * height;
    double Mse = (MseR + MseG + MseB) / 3.;
    *psnr = Mse == 0 ? 100. : 10 * log10(255 * 255 / Mse);
    return KHG_OK;
}

#ifdef KHG_TIMING
extern "C" int khg_debug_fwd(uint32_t* host, size_t words)
{
    if (words > 4 * 8192) words = 4 * 8192;
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(khg_dbg_fwd), words * 4) == hipSuccess ? 0 : KHG_ERR_HIP;
}
// Developer build only: the single-pass packer's per-entry clock stamps (eight words per entry).
extern "C" int khg_debug_packer(khg_ctx* c, uint64_t* host, size_t words)
{
    KHG_HIP(hipMemcpy(host, c->ws.chunk_off, words * 8, hipMemcpyDeviceToHost));
    return KHG_OK;
}
// Developer build only: the walk table of one workgroup of the last decode (KHG_DEC_BLOCK * KHG_MAXOFF words).
extern "C" int khg_debug_walk(khg_ctx* c, uint32_t blk, uint32_t* host)
{
    KHG_HIP(hipMemcpy(host, c->dws.walk + (size_t)blk * KHG_MAXOFF * KHG_DEC_BLOCK, (size_t)KHG_MAXOFF * KHG_DEC_BLOCK * 4,
                      hipMemcpyDeviceToHost));
    return KHG_OK;
}
// Developer build only: the per-workgroup records of the last decode (timing stamps in pad_).
extern "C" int khg_debug_blocks(khg_ctx* c, void* host, size_t bytes, uint32_t* blk0, uint32_t* nblocks)
{
    const KhgDecArgs d = khg_dec_args(c->ws, c->dws, nullptr, 0, c->max_rows, c->max_cols);
    for (int p = 0; p < 3; ++p) blk0[p] = d.blk0[p];
    *nblocks = d.nblocks;
    size_t want = (size_t)d.nblocks * sizeof(KhgBlkInfo);
    KHG_HIP(hipMemcpy(host, c->dws.blk, want < bytes ? want : bytes, hipMemcpyDeviceToHost));
    return KHG_OK;
}
#endif
