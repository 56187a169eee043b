// khg_host_tile.inl -- one image tiled by rows over several GPUs (included by khugle_hip.hip).
//
// One process and one khg_ctx per GPU; rank t of n owns image rows [rows*t/n, rows*(t+1)/n).  The library does the
// per-GPU work; the two exchanges between ranks (histogram, chunk bit counts) and the final merge of the output
// bytes are collectives the caller runs on buffers it owns (RCCL through torch.distributed in this repository's
// host layer), between the steps below.

static int khg_tile_check(khg_ctx* c, int rows, int cols, int lvl, int rank, int nranks)
{
    int rc = khg_check_image(c, rows, cols);
    if (rc) return rc;
    if (lvl < 0 || lvl > 2) return KHG_ERR_DIMS;
    KhgTile t = {rank, nranks};
    if (!khg_tile_ok(rows, t)) return KHG_ERR_DIMS;
    if ((uint64_t)rows * (uint64_t)cols >= (1ull << 31)) return KHG_ERR_RANGE;  // raster indices are exchanged as int32
    return KHG_OK;
}

extern "C" int khg_tile_input_rows(int rows, int rank, int nranks, int* row0, int* row1)
{
    KhgTile t = {rank, nranks};
    if (!row0 || !row1 || rows <= 0 || !khg_tile_ok(rows, t)) return KHG_ERR_DIMS;
    *row0 = khg_tile_in_row0(rows, t);
    *row1 = khg_tile_in_row1(rows, t);
    return KHG_OK;
}

extern "C" size_t khg_tile_chunk_count(int rows, int cols, int nranks)
{
    KhgTile t = {0, nranks};
    if (!khg_dims_ok(rows, cols) || !khg_tile_ok(rows, t)) return 0;
    KhgWorkspace none;
    memset(&none, 0, sizeof none);
    return khg_enc_args(none, rows, cols, 0, nullptr, 0, t).nchunks;
}

extern "C" int khg_tile_forward_dev(khg_ctx* c, const uint8_t* d_r, const uint8_t* d_g, const uint8_t* d_b, int rows,
                                    int cols, int lvl, int rank, int nranks, void* stream)
{
    int rc = khg_tile_check(c, rows, cols, lvl, rank, nranks);
    if (rc) return rc;
    if (!d_r || !d_g || !d_b) return KHG_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    KHG_HIP(hipSetDevice(c->device));
    const KhgTile t = {rank, nranks};
    if ((rc = khg_begin(c, st))) return rc;
    c->tile = t;
    c->tile_rows = rows; c->tile_cols = cols; c->tile_lvl = lvl;
    if ((rc = khg_enqueue_forward(c, d_r, d_g, d_b, rows, cols, lvl, nullptr, st, t))) return rc;
    return khg_end(c, st, 0);
}

// Partial (get) / image-wide (set) histogram: counts and first raster indices, 3 x 16384 words each.
extern "C" int khg_tile_hist_get(khg_ctx* c, uint32_t* d_count, uint32_t* d_first, void* stream)
{
    if (!c || !d_count || !d_first) return KHG_ERR_ARG;
    int rc = khg_begin(c, (hipStream_t)stream);
    if (rc) return rc;
    KHG_HIP(hipMemcpyAsync(d_count, c->ws.hist_cnt, 3 * KHG_NBINS * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    KHG_HIP(hipMemcpyAsync(d_first, c->ws.hist_first, 3 * KHG_NBINS * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return khg_end(c, (hipStream_t)stream, 0);
}

extern "C" int khg_tile_hist_set(khg_ctx* c, const uint32_t* d_count, const uint32_t* d_first, void* stream)
{
    if (!c || !d_count || !d_first) return KHG_ERR_ARG;
    int rc = khg_begin(c, (hipStream_t)stream);
    if (rc) return rc;
    KHG_HIP(hipMemcpyAsync(c->ws.hist_cnt, d_count, 3 * KHG_NBINS * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    KHG_HIP(hipMemcpyAsync(c->ws.hist_first, d_first, 3 * KHG_NBINS * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return khg_end(c, (hipStream_t)stream, 0);
}

// Table, tree and codes from the image-wide histogram (identical on every rank), then the bit counts of this
// rank's chunks; other ranks' entries are zero, so a sum over ranks completes the array.
extern "C" int khg_tile_lens_dev(khg_ctx* c, void* stream)
{
    if (!c || c->tile.nranks < 1) return KHG_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    KHG_HIP(hipSetDevice(c->device));
    int rc = khg_begin(c, st);
    if (rc) return rc;
    KhgTableArgs t = khg_table_args(c->ws);
    hipLaunchKernelGGL(khg_table_kernel, dim3(3), dim3(256), 0, st, t);
    KhgEncArgs e = khg_enc_args(c->ws, c->tile_rows, c->tile_cols, c->tile_lvl, nullptr, 0, c->tile);
    KHG_HIP(hipMemsetAsync(c->ws.chunk_bits, 0, (size_t)e.nchunks * 4, st));
    hipLaunchKernelGGL(khg_lens_kernel, dim3(e.nchunks < (uint32_t)KHG_TILE_GRID ? e.nchunks : (uint32_t)KHG_TILE_GRID), dim3(256), 0, st, e);
    KHG_HIP(hipGetLastError());
    return khg_end(c, st, 0);
}

extern "C" int khg_tile_bits_get(khg_ctx* c, uint32_t* d_bits, void* stream)
{
    if (!c || !d_bits || c->tile.nranks < 1) return KHG_ERR_ARG;
    size_t n = khg_tile_chunk_count(c->tile_rows, c->tile_cols, c->tile.nranks);
    int rc = khg_begin(c, (hipStream_t)stream);
    if (rc) return rc;
    KHG_HIP(hipMemcpyAsync(d_bits, c->ws.chunk_bits, n * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return khg_end(c, (hipStream_t)stream, 0);
}

extern "C" int khg_tile_chunk_off_get(khg_ctx* c, uint64_t* d_off, void* stream)
{
    if (!c || !d_off || c->tile.nranks < 1) return KHG_ERR_ARG;
    size_t n = khg_tile_chunk_count(c->tile_rows, c->tile_cols, c->tile.nranks);
    int rc = khg_begin(c, (hipStream_t)stream);
    if (rc) return rc;
    KHG_HIP(hipMemcpyAsync(d_off, c->ws.chunk_off, n * 8, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return khg_end(c, (hipStream_t)stream, 0);
}

extern "C" int khg_tile_bits_set(khg_ctx* c, const uint32_t* d_bits, void* stream)
{
    if (!c || !d_bits || c->tile.nranks < 1) return KHG_ERR_ARG;
    size_t n = khg_tile_chunk_count(c->tile_rows, c->tile_cols, c->tile.nranks);
    int rc = khg_begin(c, (hipStream_t)stream);
    if (rc) return rc;
    KHG_HIP(hipMemcpyAsync(c->ws.chunk_bits, d_bits, n * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return khg_end(c, (hipStream_t)stream, 0);
}

// Offsets of all chunks (from the completed bit counts), then this rank's share of the file: its chunks' bits at
// their final positions, rank 0 also the header and the tables, zero everywhere else.  The shares of all ranks
// have no set bit in common, so adding them as 32-bit words (or OR-ing them) gives the file.
// khg_encode_result returns the status and the length of the whole file on every rank.
extern "C" int khg_tile_pack_dev(khg_ctx* c, uint8_t* d_out, size_t cap, void* stream)
{
    if (!c || !d_out || ((uintptr_t)d_out & 3) || c->tile.nranks < 1) return KHG_ERR_ARG;
    if (cap < 40) return KHG_ERR_CAP;
    hipStream_t st = (hipStream_t)stream;
    KHG_HIP(hipSetDevice(c->device));
    int rc = khg_begin(c, st);
    if (rc) return rc;
    KhgEncArgs e = khg_enc_args(c->ws, c->tile_rows, c->tile_cols, c->tile_lvl, d_out, cap, c->tile);
    // offsets and the file's length first, so that only the bytes the file will occupy are zeroed (the capacity is
    // sized for 15 bits per symbol); the header and the tables follow the streams (rank 0 only)
    KhgEncArgs scan = e;
    scan.write_meta = 0;
    hipLaunchKernelGGL(khg_scan_kernel, dim3(1), dim3(1024), 0, st, scan);
    hipLaunchKernelGGL(khg_clear_share_kernel, dim3(2048), dim3(256), 0, st, d_out, cap, (const KhgResult*)c->ws.res);
    hipLaunchKernelGGL(khg_pack_kernel, dim3(e.nchunks < (uint32_t)KHG_TILE_GRID ? e.nchunks : (uint32_t)KHG_TILE_GRID), dim3(256), 0, st, e);
    if (e.write_meta) hipLaunchKernelGGL(khg_meta_kernel, dim3(1), dim3(1024), 0, st, e);
    KHG_HIP(hipGetLastError());
    KHG_HIP(hipMemcpyAsync(c->h_res, c->ws.res, sizeof(KhgResult), hipMemcpyDeviceToHost, st));
    return khg_end(c, st, 1);
}

// Decode of a tiled image: every rank decodes the whole entropy-coded stream (it cannot be split before it is
// decoded) and reconstructs its own rows; d_r/d_g/d_b receive rows [rows*rank/nranks, rows*(rank+1)/nranks).
extern "C" int khg_tile_decode_dev(khg_ctx* c, const uint8_t* d_comp, size_t len, int rows, int cols, int rank,
                                   int nranks, uint8_t* d_r, uint8_t* d_g, uint8_t* d_b, void* stream)
{
    int rc = khg_tile_check(c, rows, cols, 0, rank, nranks);
    if (rc) return rc;
    const KhgTile t = {rank, nranks};
    return khg_decode_enqueue(c, d_comp, len, rows, cols, d_r, d_g, d_b, stream, t);
}

// ---- tiled decode with the entropy decode split over the ranks.  Every rank holds the whole file; rank t maps and
// decodes the workgroup spans [nb * t / n, nb * (t + 1) / n) of each plane's stream.  Between the steps the caller
//   1. sums the workgroup records over the ranks (khg_tile_blk_get / all-reduce SUM / khg_tile_blk_set: the
//      records of other ranks' spans are zero here), so that every rank can chain all spans from bit 0;
//   2. moves decoded symbols from the ranks that hold them to the ranks whose rows need them
//      (khg_tile_sym_get / point-to-point / khg_tile_sym_set, ranges from khg_tile_decode_write_dev and
//      khg_tile_need_ranges).

static KhgDecArgs khg_tdec_args(khg_ctx* c)
{
    KhgDecArgs d = khg_dec_args(c->ws, c->dws, c->tdec_comp, c->tdec_len, c->tile_rows, c->tile_cols);
    d.part_rank = (uint32_t)c->tile.rank;
    d.part_nranks = (uint32_t)c->tile.nranks;
    return d;
}

static unsigned khg_tdec_grid(const khg_ctx* c)
{
    unsigned g = 0;
    for (int p = 0; p < 3; ++p)
        g += khg_part_begin(c->tdec_nb[p], (uint32_t)c->tile.rank + 1, (uint32_t)c->tile.nranks) -
             khg_part_begin(c->tdec_nb[p], (uint32_t)c->tile.rank, (uint32_t)c->tile.nranks);
    return g;
}

extern "C" int khg_tile_decode_map_dev(khg_ctx* c, const uint8_t* d_comp, size_t len, int rows, int cols, int rank,
                                       int nranks, void* stream)
{
    int rc = khg_tile_check(c, rows, cols, 0, rank, nranks);
    if (rc) return rc;
    if (!d_comp || ((uintptr_t)d_comp & 3)) return KHG_ERR_ARG;
    if (len < 36 + 24) return KHG_ERR_FORMAT;
    hipStream_t st = (hipStream_t)stream;
    KHG_HIP(hipSetDevice(c->device));
    if ((rc = khg_begin(c, st))) return rc;
    uint8_t hdr36[36];  // the host needs the stream lengths to size the launches and the exchange
    KHG_HIP(hipMemcpyAsync(hdr36, d_comp, 36, hipMemcpyDeviceToHost, st));
    KHG_HIP(hipStreamSynchronize(st));
    c->tile = KhgTile{rank, nranks};
    c->tile_rows = rows; c->tile_cols = cols;
    c->tdec_comp = d_comp; c->tdec_len = len;
    khg_tile_active_blocks(hdr36, c->tdec_nb);
    const KhgDecArgs d = khg_tdec_args(c);
    for (int p = 0; p < 3; ++p) {  // a stream can span no more workgroups than the workspace holds
        const uint32_t cap = (p < 2 ? d.blk0[p + 1] : d.nblocks) - d.blk0[p];
        if (c->tdec_nb[p] > cap) c->tdec_nb[p] = cap;  // the device-side header check fails such a file
    }
    for (int p = 0; p < 3; ++p)
        if (c->tdec_nb[p])
            KHG_HIP(hipMemsetAsync(c->dws.blk + d.blk0[p], 0, (size_t)c->tdec_nb[p] * sizeof(KhgBlkInfo), st));
    hipLaunchKernelGGL(khg_dec_header_kernel, dim3(1), dim3(256), 0, st, d);
    hipLaunchKernelGGL(khg_dec_tree_kernel, dim3(3), dim3(256), 0, st, d);
    const unsigned grid = khg_tdec_grid(c);
    if (grid) {
        hipLaunchKernelGGL((khg_dec_map_kernel<KHG_MAP_SMALL>), dim3(grid), dim3(KHG_MAP_THREADS), 0, st, d);
        hipLaunchKernelGGL((khg_dec_map_kernel<KHG_MAXOFF>), dim3(grid), dim3(KHG_MAP_THREADS), 0, st, d);
    }
    KHG_HIP(hipGetLastError());
    c->tdec_step = 1;
    return khg_end(c, st, 0);
}

// 32-bit words of the workgroup records of all three planes (what khg_tile_blk_get / _set move).
extern "C" size_t khg_tile_blk_words(khg_ctx* c)
{
    if (!c || c->tdec_step < 1) return 0;
    return ((size_t)c->tdec_nb[0] + c->tdec_nb[1] + c->tdec_nb[2]) * (sizeof(KhgBlkInfo) / 4);
}

static int khg_tile_blk_copy(khg_ctx* c, uint32_t* d_words, void* stream, bool get)
{
    if (!c || !d_words || c->tdec_step < 1) return KHG_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    KHG_HIP(hipSetDevice(c->device));
    int rc = khg_begin(c, st);
    if (rc) return rc;
    const KhgDecArgs d = khg_tdec_args(c);
    size_t at = 0;
    for (int p = 0; p < 3; ++p) {
        const size_t bytes = (size_t)c->tdec_nb[p] * sizeof(KhgBlkInfo);
        if (!bytes) continue;
        void* rec = c->dws.blk + d.blk0[p];
        KHG_HIP(hipMemcpyAsync(get ? (void*)(d_words + at) : rec, get ? rec : (void*)(d_words + at), bytes,
                               hipMemcpyDeviceToDevice, st));
        at += bytes / 4;
    }
    return khg_end(c, st, 0);
}

extern "C" int khg_tile_blk_get(khg_ctx* c, uint32_t* d_words, void* stream) { return khg_tile_blk_copy(c, d_words, stream, true); }
extern "C" int khg_tile_blk_set(khg_ctx* c, const uint32_t* d_words, void* stream)
{
    return khg_tile_blk_copy(c, (uint32_t*)d_words, stream, false);
}

// Chains all spans (every rank, from the summed records), decodes this rank's spans into its symbol planes and
// reports where each rank's symbols start: bounds[p * (nranks + 1) + t] = raster index of the first symbol rank t
// decodes in plane p, bounds[p * (nranks + 1) + nranks] = symbols decoded in the plane (the cells behind them are
// zero on every rank, Huffman.h:35).  Waits for the stream.
extern "C" int khg_tile_decode_write_dev(khg_ctx* c, uint32_t* bounds, void* stream)
{
    if (!c || !bounds || c->tdec_step != 1) return KHG_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    KHG_HIP(hipSetDevice(c->device));
    int rc = khg_begin(c, st);
    if (rc) return rc;
    const KhgDecArgs d = khg_tdec_args(c);
    const int n = c->tile.nranks;
    khg_launch_chain(st, d);
    const unsigned grid = khg_tdec_grid(c);
    if (grid) hipLaunchKernelGGL(khg_dec_write_kernel, dim3(grid), dim3(256), 0, st, d);
    KHG_HIP(hipGetLastError());
    KHG_HIP(hipMemcpyAsync(c->h_res, c->ws.res, sizeof(KhgResult), hipMemcpyDeviceToHost, st));
    for (int p = 0; p < 3; ++p)
        for (int t = 0; t < n; ++t) {
            const uint32_t b = khg_part_begin(c->tdec_nb[p], (uint32_t)t, (uint32_t)n);
            uint32_t* dst = &bounds[p * (n + 1) + t];
            if (b < c->tdec_nb[p]) KHG_HIP(hipMemcpyAsync(dst, &c->dws.blk[d.blk0[p] + b].out, 4, hipMemcpyDeviceToHost, st));
            else *dst = 0xFFFFFFFFu;  // no span: patched below
        }
    KHG_HIP(hipStreamSynchronize(st));
    for (int p = 0; p < 3; ++p) {
        uint32_t total = (uint32_t)c->h_res->bits[p];  // symbols decoded (khg_dec_chain_body)
        if (total > d.n[p]) total = d.n[p];
        bounds[p * (n + 1) + n] = total;
        for (int t = n - 1; t >= 0; --t) {  // ranks without a span own nothing; starts never pass the total
            uint32_t& v = bounds[p * (n + 1) + t];
            if (v > bounds[p * (n + 1) + t + 1]) v = bounds[p * (n + 1) + t + 1];
        }
    }
    c->tdec_step = 2;
    return khg_end(c, st, 0);
}

extern "C" int khg_tile_need_ranges(int rows, int cols, int rank, int nranks, int plane, uint32_t* ranges)
{
    KhgTile t = {rank, nranks};
    if (!ranges || plane < 0 || plane > 2 || !khg_dims_ok(rows, cols) || !khg_tile_ok(rows, t)) return KHG_ERR_DIMS;
    uint32_t r[4][2];
    const int n = khg_tile_need_ranges_impl(rows, cols, t, plane, r);
    for (int i = 0; i < n; ++i) {
        ranges[2 * i] = r[i][0];
        ranges[2 * i + 1] = r[i][1];
    }
    return n;
}

static int khg_tile_sym_copy(khg_ctx* c, int plane, uint32_t first, uint32_t count, int16_t* d_buf, void* stream, bool get)
{
    if (!c || !d_buf || plane < 0 || plane > 2 || c->tdec_step != 2) return KHG_ERR_ARG;
    const uint64_t n = (uint64_t)c->tile_rows * c->tile_cols / (plane ? 4 : 1);
    if ((uint64_t)first + count > n) return KHG_ERR_ARG;
    if (!count) return KHG_OK;
    hipStream_t st = (hipStream_t)stream;
    KHG_HIP(hipSetDevice(c->device));
    int rc = khg_begin(c, st);
    if (rc) return rc;
    int16_t* sym = c->ws.sym[plane] + first;
    KHG_HIP(hipMemcpyAsync(get ? d_buf : sym, get ? sym : d_buf, (size_t)count * 2, hipMemcpyDeviceToDevice, st));
    return khg_end(c, st, 0);
}

extern "C" int khg_tile_sym_get(khg_ctx* c, int plane, uint32_t first, uint32_t count, int16_t* d_dst, void* stream)
{
    return khg_tile_sym_copy(c, plane, first, count, d_dst, stream, true);
}
extern "C" int khg_tile_sym_set(khg_ctx* c, int plane, uint32_t first, uint32_t count, const int16_t* d_src, void* stream)
{
    return khg_tile_sym_copy(c, plane, first, count, (int16_t*)d_src, stream, false);
}

// The inverse transform and colour conversion of this rank's rows from the symbols now in place; result and status
// as after khg_tile_decode_dev (khg_last_status).
extern "C" int khg_tile_decode_inverse_dev(khg_ctx* c, uint8_t* d_r, uint8_t* d_g, uint8_t* d_b, void* stream)
{
    if (!c || c->tdec_step != 2) return KHG_ERR_ARG;
    if (!d_r || !d_g || !d_b || ((uintptr_t)d_r & 7) || ((uintptr_t)d_g & 7) || ((uintptr_t)d_b & 7)) return KHG_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    KHG_HIP(hipSetDevice(c->device));
    int rc = khg_begin(c, st);
    if (rc) return rc;
    khg_enqueue_inverse(c, c->tile_rows, c->tile_cols, d_r, d_g, d_b, st, c->tile, nullptr);
    KHG_HIP(hipGetLastError());
    KHG_HIP(hipMemcpyAsync(c->h_res, c->ws.res, sizeof(KhgResult), hipMemcpyDeviceToHost, st));
    c->tdec_step = 0;
    return khg_end(c, st, 2);
}
