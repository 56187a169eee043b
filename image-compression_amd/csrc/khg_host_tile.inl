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
    hipLaunchKernelGGL(khg_lens_kernel, dim3(e.nchunks), dim3(256), 0, st, e);
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
    KHG_HIP(hipMemsetAsync(d_out, 0, cap & ~(size_t)3, st));
    hipLaunchKernelGGL(khg_scan_kernel, dim3(1), dim3(1024), 0, st, e);
    hipLaunchKernelGGL(khg_pack_kernel, dim3(e.nchunks), dim3(256), 0, st, e);
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
