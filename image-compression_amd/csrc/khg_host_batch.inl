// khg_host_batch.inl -- batches of frames held in host memory (included by khugle_hip.hip).
//
// The counterpart of calling LoadBMPAndCompress / LoadComp (Main.cpp:180-334) in a loop: frames are independent, so
// `depth` contexts work on consecutive frames, each on its own stream with its own pinned staging buffers, and the
// copies of one frame overlap the kernels of the others:
//     host frame -> (pinned staging) -> H2D -> encode/decode kernels -> D2H -> (pinned staging) -> host result
// Caller buffers that are already pinned (hipHostMalloc / hipHostRegister) are used directly.

struct khg_batch_slot {
    khg_ctx* ctx;
    hipStream_t stream;
    hipEvent_t copied;     // the slot's last D2H has finished
    uint8_t* d_rgb;        // 3 planes
    uint8_t* d_comp;
    uint8_t* h_rgb;        // pinned staging
    uint8_t* h_comp;
    int frame;             // frame in flight, -1 if none
};

struct khg_batch {
    int device, max_rows, max_cols, depth;
    size_t comp_cap;
    khg_batch_slot* slot;
};

static bool khg_is_pinned(const void* p)
{
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) {
        (void)hipGetLastError();  // plain pageable memory: not an error for the caller
        return false;
    }
    return at.type == hipMemoryTypeHost;
}

extern "C" void khg_batch_destroy(khg_batch* b)
{
    if (!b) return;
    (void)hipSetDevice(b->device);
    for (int i = 0; b->slot && i < b->depth; ++i) {
        khg_batch_slot& s = b->slot[i];
        if (s.stream) (void)hipStreamSynchronize(s.stream);
        if (s.ctx) khg_ctx_destroy(s.ctx);
        if (s.d_rgb) (void)hipFree(s.d_rgb);
        if (s.d_comp) (void)hipFree(s.d_comp);
        if (s.h_rgb) (void)hipHostFree(s.h_rgb);
        if (s.h_comp) (void)hipHostFree(s.h_comp);
        if (s.copied) (void)hipEventDestroy(s.copied);
        if (s.stream) (void)hipStreamDestroy(s.stream);
    }
    delete[] b->slot;
    delete b;
}

extern "C" int khg_batch_create(int device, int max_rows, int max_cols, int depth, khg_batch** out)
{
    if (!out || depth < 1 || depth > 64 || !khg_dims_ok(max_rows, max_cols)) return KHG_ERR_ARG;
    *out = nullptr;
    khg_batch* b = new (std::nothrow) khg_batch;
    if (!b) return KHG_ERR_NOMEM;
    memset(b, 0, sizeof *b);
    b->device = device; b->max_rows = max_rows; b->max_cols = max_cols; b->depth = depth;
    b->comp_cap = khg_comp_bound_impl(max_rows, max_cols);
    b->slot = new (std::nothrow) khg_batch_slot[depth];
    if (!b->slot) { delete b; return KHG_ERR_NOMEM; }
    memset(b->slot, 0, sizeof(khg_batch_slot) * depth);
    const size_t n3 = 3 * (size_t)max_rows * max_cols;
    for (int i = 0; i < depth; ++i) {
        khg_batch_slot& s = b->slot[i];
        s.frame = -1;
        int rc = khg_ctx_create(device, max_rows, max_cols, &s.ctx);
        if (rc == KHG_OK && (hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) != hipSuccess ||
                             hipEventCreateWithFlags(&s.copied, hipEventDisableTiming) != hipSuccess))
            rc = KHG_ERR_HIP;
        if (rc == KHG_OK && (hipMalloc((void**)&s.d_rgb, n3) != hipSuccess || hipMalloc((void**)&s.d_comp, b->comp_cap) != hipSuccess ||
                             hipHostMalloc((void**)&s.h_rgb, n3) != hipSuccess ||
                             hipHostMalloc((void**)&s.h_comp, b->comp_cap) != hipSuccess))
            rc = KHG_ERR_NOMEM;
        if (rc != KHG_OK) {
            khg_batch_destroy(b);
            return rc;
        }
    }
    *out = b;
    return KHG_OK;
}

// Finishes the encode in flight on a slot: waits, then hands the bytes to the caller's buffer.
static void khg_batch_finish_encode(khg_batch* b, khg_batch_slot& s, uint8_t* const* out, const size_t* cap,
                                    size_t* out_len, int* status)
{
    const int f = s.frame;
    if (f < 0) return;
    s.frame = -1;
    size_t len = 0;
    int rc = khg_encode_result(s.ctx, &len);
    if (rc == KHG_OK && len > cap[f]) rc = KHG_ERR_CAP;
    if (rc == KHG_OK) {
        // the file is small next to the frame: one synchronous copy, straight into a pinned caller buffer
        // or through the staging buffer
        if (khg_is_pinned(out[f])) {
            if (hipMemcpyAsync(out[f], s.d_comp, len, hipMemcpyDeviceToHost, s.stream) != hipSuccess ||
                hipStreamSynchronize(s.stream) != hipSuccess)
                rc = KHG_ERR_HIP;
        } else if (hipMemcpyAsync(s.h_comp, s.d_comp, len, hipMemcpyDeviceToHost, s.stream) != hipSuccess ||
                   hipStreamSynchronize(s.stream) != hipSuccess) {
            rc = KHG_ERR_HIP;
        } else {
            memcpy(out[f], s.h_comp, len);
        }
    }
    out_len[f] = rc == KHG_OK ? len : 0;
    status[f] = rc;
}

extern "C" int khg_batch_encode(khg_batch* b, int n, const uint8_t* const* rgb, int rows, int cols, int lvl,
                                uint8_t* const* out, const size_t* cap, size_t* out_len, int* status)
{
    if (!b || n < 0 || !rgb || !out || !cap || !out_len || !status) return KHG_ERR_ARG;
    if (!khg_dims_ok(rows, cols) || rows > b->max_rows || cols > b->max_cols || lvl < 0 || lvl > 2) return KHG_ERR_DIMS;
    KHG_HIP(hipSetDevice(b->device));
    const size_t np = (size_t)rows * cols;
    for (int f = 0; f < n; ++f) {
        khg_batch_slot& s = b->slot[f % b->depth];
        khg_batch_finish_encode(b, s, out, cap, out_len, status);
        if (!rgb[f] || !out[f]) {
            status[f] = KHG_ERR_ARG;
            out_len[f] = 0;
            continue;
        }
        const uint8_t* src = rgb[f];
        if (!khg_is_pinned(src)) {  // pageable: through the slot's pinned buffer (free again: its H2D was waited for)
            memcpy(s.h_rgb, src, 3 * np);
            src = s.h_rgb;
        }
        int rc = KHG_OK;
        if (hipMemcpyAsync(s.d_rgb, src, 3 * np, hipMemcpyHostToDevice, s.stream) != hipSuccess) rc = KHG_ERR_HIP;
        if (rc == KHG_OK)
            rc = khg_encode_dev(s.ctx, s.d_rgb, s.d_rgb + np, s.d_rgb + 2 * np, rows, cols, lvl, s.d_comp, b->comp_cap, s.stream);
        if (rc != KHG_OK) {
            (void)hipStreamSynchronize(s.stream);  // the H2D copy may still be reading the slot's staging buffer
            status[f] = rc;
            out_len[f] = 0;
            continue;
        }
        s.frame = f;
    }
    for (int i = 0; i < b->depth; ++i) khg_batch_finish_encode(b, b->slot[i], out, cap, out_len, status);
    for (int f = 0; f < n; ++f)
        if (status[f] != KHG_OK) return status[f];
    return KHG_OK;
}

static void khg_batch_finish_decode(khg_batch* b, khg_batch_slot& s, uint8_t* const* rgb_out, const int* rows,
                                    const int* cols, int* status)
{
    const int f = s.frame;
    if (f < 0) return;
    s.frame = -1;
    int rc = khg_last_status(s.ctx);  // kernels done
    // the D2H copy was enqueued whatever the decode's status turns out to be: it must have finished before the
    // slot's buffers (or the caller's) are touched again
    if (hipEventSynchronize(s.copied) != hipSuccess && rc == KHG_OK) rc = KHG_ERR_HIP;
    if (rc == KHG_OK && !khg_is_pinned(rgb_out[f])) memcpy(rgb_out[f], s.h_rgb, 3 * (size_t)rows[f] * cols[f]);
    status[f] = rc;
}

// rows[f] x cols[f] is read from each file's header (khg_peek); rgb_out[f] must hold 3 planes of that size
// (khg_peek gives the sizes beforehand).
extern "C" int khg_batch_decode(khg_batch* b, int n, const uint8_t* const* comp, const size_t* len,
                                uint8_t* const* rgb_out, int* status)
{
    if (!b || n < 0 || !comp || !len || !rgb_out || !status) return KHG_ERR_ARG;
    KHG_HIP(hipSetDevice(b->device));
    int* rows = new (std::nothrow) int[2 * (size_t)(n > 0 ? n : 1)];
    if (!rows) return KHG_ERR_NOMEM;
    int* cols = rows + (n > 0 ? n : 1);
    for (int f = 0; f < n; ++f) {
        khg_batch_slot& s = b->slot[f % b->depth];
        khg_batch_finish_decode(b, s, rgb_out, rows, cols, status);
        khg_header hd;
        int rc = (comp[f] && rgb_out[f]) ? khg_peek(comp[f], len[f], &hd) : KHG_ERR_ARG;
        if (rc == KHG_OK && (hd.h > b->max_rows || hd.w > b->max_cols)) rc = KHG_ERR_DIMS;
        if (rc == KHG_OK && ((len[f] + 3) & ~(size_t)3) > b->comp_cap) rc = KHG_ERR_CAP;
        if (rc != KHG_OK) {
            status[f] = rc;
            continue;
        }
        rows[f] = hd.h; cols[f] = hd.w;
        const size_t np = (size_t)hd.h * hd.w;
        const uint8_t* src = comp[f];
        if (!khg_is_pinned(src)) {
            memcpy(s.h_comp, src, len[f]);
            src = s.h_comp;
        }
        if (hipMemcpyAsync(s.d_comp, src, len[f], hipMemcpyHostToDevice, s.stream) != hipSuccess) rc = KHG_ERR_HIP;
        if (rc == KHG_OK)
            rc = khg_decode_dev(s.ctx, s.d_comp, len[f], hd.h, hd.w, s.d_rgb, s.d_rgb + np, s.d_rgb + 2 * np, s.stream);
        if (rc == KHG_OK) {
            uint8_t* dst = khg_is_pinned(rgb_out[f]) ? rgb_out[f] : s.h_rgb;
            if (hipMemcpyAsync(dst, s.d_rgb, 3 * np, hipMemcpyDeviceToHost, s.stream) != hipSuccess ||
                hipEventRecord(s.copied, s.stream) != hipSuccess)
                rc = KHG_ERR_HIP;
        }
        if (rc != KHG_OK) {
            // whatever was enqueued before the failure (the H2D copy, the decode) may still use the slot's buffers
            (void)hipStreamSynchronize(s.stream);
            (void)khg_last_status(s.ctx);  // and a decode that was enqueued leaves the context waiting for this call
            status[f] = rc;
            continue;
        }
        s.frame = f;
    }
    for (int i = 0; i < b->depth; ++i) khg_batch_finish_decode(b, b->slot[i], rgb_out, rows, cols, status);
    delete[] rows;
    for (int f = 0; f < n; ++f)
        if (status[f] != KHG_OK) return status[f];
    return KHG_OK;
}
