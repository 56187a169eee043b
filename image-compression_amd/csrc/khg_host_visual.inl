// khg_host_visual.inl -- host side of the pictures and status lines the application shows beside a result
// (included by khugle_hip.hip).

static inline unsigned khg_vis_grid(int rows, int cols) { return (unsigned)(((size_t)rows * cols + 255) / 256); }

extern "C" int khg_status_strings(int rows, int cols, int lvl, int k_y, int k_cb, int k_cr, char* step_str, size_t step_cap,
                                  char* rle_str, size_t rle_cap)
{
    if (!khg_dims_ok(rows, cols) || lvl < 0 || lvl > 2) return KHG_ERR_DIMS;
    if (step_str) {  // Compress.h:294-296, Decompress.h:265-267
        KhgQuant q;
        khg_make_quant(rows, cols, lvl, 0, &q);
        double lo, hi;
        khg_vis_step_range(rows, cols, lvl, q, &lo, &hi);
        if (snprintf(step_str, step_cap, "Step Size (min, max) : %.3f, %.3f", lo, hi) >= (int)step_cap) return KHG_ERR_CAP;
    }
    if (rle_str) {  // Compress.h:299-300, Decompress.h:270-271
        if (snprintf(rle_str, rle_cap, "RLE Size (Y, Cb, Cr) : %d, %d, %d", k_y, k_cb, k_cr) >= (int)rle_cap) return KHG_ERR_CAP;
    }
    return KHG_OK;
}

extern "C" int khg_vis_steps_dev(int rows, int cols, int lvl, uint8_t* d_r, uint8_t* d_g, uint8_t* d_b, void* stream)
{
    if (!khg_dims_ok(rows, cols) || lvl < 0 || lvl > 2) return KHG_ERR_DIMS;
    if (!d_r || !d_g || !d_b) return KHG_ERR_ARG;
    KhgVisStepArgs a;
    a.rows = rows; a.cols = cols; a.lvl = lvl;
    khg_make_quant(rows, cols, lvl, 0, &a.q);
    khg_vis_step_range(rows, cols, lvl, a.q, &a.lo, &a.hi);
    a.r = d_r; a.g = d_g; a.b = d_b;
    hipLaunchKernelGGL(khg_vis_step_kernel, dim3(khg_vis_grid(rows, cols)), dim3(256), 0, (hipStream_t)stream, a);
    KHG_HIP(hipGetLastError());
    return KHG_OK;
}

extern "C" int khg_vis_bars_dev(int rows, int cols, const long long* len_bytes, uint8_t* d_r, uint8_t* d_g, uint8_t* d_b,
                                void* stream)
{
    if (!khg_dims_ok(rows, cols)) return KHG_ERR_DIMS;
    if (!len_bytes || !d_r || !d_g || !d_b) return KHG_ERR_ARG;
    KhgVisBarArgs a;
    a.rows = rows; a.cols = cols;
    for (int c = 0; c < 3; ++c) {
        if (len_bytes[c] < 0) return KHG_ERR_ARG;
        a.len[c] = len_bytes[c];
    }
    a.r = d_r; a.g = d_g; a.b = d_b;
    hipLaunchKernelGGL(khg_vis_bar_kernel, dim3(khg_vis_grid(rows, cols)), dim3(256), 0, (hipStream_t)stream, a);
    KHG_HIP(hipGetLastError());
    return KHG_OK;
}

extern "C" int khg_vis_coeffs_dev(khg_ctx* c, const uint8_t* d_r, const uint8_t* d_g, const uint8_t* d_b, int rows, int cols,
                                  uint8_t* d_or, uint8_t* d_og, uint8_t* d_ob, void* stream)
{
    int rc = khg_check_image(c, rows, cols);
    if (rc) return rc;
    if (!d_r || !d_g || !d_b || !d_or || !d_og || !d_ob) return KHG_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    KHG_HIP(hipSetDevice(c->device));
    if ((rc = khg_stage_f64(c))) return rc;
    if ((rc = khg_begin(c, st))) return rc;
    const size_t n = (size_t)rows * cols;
    double* coef[3] = {c->stage_f64, c->stage_f64 + n, c->stage_f64 + n + n / 4};
    if ((rc = khg_enqueue_forward(c, d_r, d_g, d_b, rows, cols, 0, coef, st))) return rc;  // Compress.h:108-114
    KhgVisYccArgs a;
    memset(&a, 0, sizeof a);
    a.rows = rows; a.cols = cols;
    a.y = coef[0]; a.cb = coef[1]; a.cr = coef[2];
    a.r = d_or; a.g = d_og; a.b = d_ob;
    hipLaunchKernelGGL(khg_vis_ycc_kernel, dim3(khg_vis_grid(rows, cols)), dim3(256), 0, st, a);
    KHG_HIP(hipGetLastError());
    return khg_end(c, st, 0);
}

extern "C" int khg_vis_dequant_dev(khg_ctx* c, const uint8_t* d_comp, size_t len, int rows, int cols, uint8_t* d_or,
                                   uint8_t* d_og, uint8_t* d_ob, void* stream)
{
    if (!d_or || !d_og || !d_ob) return KHG_ERR_ARG;
    return khg_decode_enqueue(c, d_comp, len, rows, cols, d_or, d_og, d_ob, stream, KhgTile{0, 1}, nullptr, 0, true);
}

extern "C" int khg_downsample_dev(const uint8_t* d_src, int src_rows, int src_cols, uint8_t* d_dst, int dst_rows, int dst_cols,
                                  void* stream)
{
    if (!d_src || !d_dst) return KHG_ERR_ARG;
    if (src_rows < 1 || src_cols < 1 || dst_rows < 1 || dst_cols < 1) return KHG_ERR_DIMS;
    // single-precision positions: (dst - 1) * (src / dst) stays below src only while 1 / dst is well above the
    // rounding error, so sides are limited to 2^22 (the application asks for 256)
    if (src_rows > (1 << 22) || src_cols > (1 << 22) || dst_rows > (1 << 22) || dst_cols > (1 << 22)) return KHG_ERR_DIMS;
    KhgDownArgs a;
    a.src = d_src; a.src_h = src_rows; a.src_w = src_cols;
    a.dst = d_dst; a.dst_h = dst_rows; a.dst_w = dst_cols;
    hipLaunchKernelGGL(khg_downsample_kernel, dim3(khg_vis_grid(dst_rows, dst_cols)), dim3(256), 0, (hipStream_t)stream, a);
    KHG_HIP(hipGetLastError());
    return KHG_OK;
}
