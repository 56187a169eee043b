/*
 * khugle_hip.h -- C ABI of the MI355X (gfx950) implementation of the KhuGle codec path.
 *
 * The reference has no FFI: its interface for this path is a set of inline C++
 * functions called from Main.cpp on row-pointer matrices.  Each entry point below
 * names the reference interface it stands in for.  Plain pointers and sizes only.
 *
 * Conventions
 *   - return 0 (KHG_OK) on success, a negative khg_status otherwise; nothing throws.
 *   - images are planar u8: three pointers to rows*cols bytes each, row 0 on top,
 *     the layout CKhuGleSignal::m_Red/m_Green/m_Blue hold after ReadBmp
 *     (KhuGleSignal.cpp:264-272) flattened row-major.
 *   - rows and cols must be positive multiples of 8 (every transform length even at
 *     both DWT levels of luma and chroma; the reference silently requires the same).
 *   - "_dev" entry points take DEVICE pointers, enqueue on `stream` (a hipStream_t,
 *     NULL = default stream) and do not synchronise; results are read with
 *     khg_encode_result / khg_last_status.  The others take HOST pointers and are
 *     synchronous.  A context is bound to one device and owns one workspace, so it runs one job
 *     at a time: after khg_encode_dev / khg_decode_dev (and their ycbcr / tile forms) every entry
 *     point that would enqueue more work on the context returns KHG_ERR_BUSY until the result has
 *     been fetched (khg_encode_result / khg_last_status); work enqueued on a different stream than
 *     the context's previous work is ordered behind it on the device.  A context is not thread
 *     safe: use one context per host thread / per frame in flight.
 *   - no entry point falls back to the CPU: without a usable HIP device every call
 *     returns KHG_ERR_HIP.
 */
#ifndef KHUGLE_HIP_H
#define KHUGLE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum khg_status {
    KHG_OK = 0,
    KHG_ERR_DIMS = -1,      /* rows/cols not positive multiples of 8, above the context's limits, or lvl not 0..2 */
    KHG_ERR_CAP = -2,       /* output buffer too small (khg_comp_bound gives a safe size) */
    KHG_ERR_FORMAT = -3,    /* malformed .comp (lengths, pads, table sizes inconsistent with the buffer) */
    KHG_ERR_RANGE = -4,     /* a plane's stream reaches 2^31 bits (header fields are int, Type.h:5),
                               a symbol leaves [-8192, 8191], or a code exceeds 56 bits */
    KHG_ERR_NOMEM = -5,
    KHG_ERR_HIP = -6,       /* no device / HIP runtime error */
    KHG_ERR_ARG = -7,
    KHG_ERR_BUSY = -8       /* the context holds a job whose result has not been fetched yet */
} khg_status;

typedef struct khg_ctx khg_ctx;

/* 36-byte header of a .comp file: HeaderInfo, Type.h:3-28, as laid out by WriteHeader, Writer.h:9-23. */
typedef struct khg_header {
    int32_t y_len, cb_len, cr_len;             /* padded bit counts */
    int32_t rle_y_len, rle_cb_len, rle_cr_len; /* symbol-table entry counts */
    int32_t h, w;                              /* file offsets 24 and 28: rows, cols */
    uint8_t y_pad, cb_pad, cr_pad, lvl;
} khg_header;

/* Workspace for images up to max_rows x max_cols on HIP device `device`. */
int khg_ctx_create(int device, int max_rows, int max_cols, khg_ctx **out);
void khg_ctx_destroy(khg_ctx *ctx);
const char *khg_status_string(int status);

/* Upper bound of the .comp size for a rows x cols image (multiple of 4). */
size_t khg_comp_bound(int rows, int cols);

/* ---- encode: LoadBMPAndCompress's codec part (Main.cpp:196-208): u8 -> double copy,
 * RGB2YCbCr (Colorspace.h:5-23), CompressImage (Compress.h:104-408, without its GUI
 * visualisation images) and the byte layout WriteAll (Writer.h:43-69) puts on disk.
 * `out` receives exactly the bytes of the .comp file.  lvl: 0 Normal, 1 High, 2 Extreme. */
int khg_encode(khg_ctx *ctx, const uint8_t *r, const uint8_t *g, const uint8_t *b, int rows, int cols, int lvl,
               uint8_t *out, size_t cap, size_t *out_len);
int khg_encode_dev(khg_ctx *ctx, const uint8_t *d_r, const uint8_t *d_g, const uint8_t *d_b, int rows, int cols,
                   int lvl, uint8_t *d_out, size_t cap, void *stream);
/* The same from the pixel array of a 24-bit BMP as it lies in the file: B, G, R bytes per pixel, rows bottom-up,
 * `stride` bytes per row (3 * cols rounded up to a multiple of 4; stride * rows < 2^32).  Replaces ReadBmp's
 * conversion loop (KhuGleSignal.cpp:264-272: m_Red[y][x] = pixel[(h - 1 - y) * stride + 3 x + 2], ...) followed by
 * the u8 -> double copy of Main.cpp:196-204: the kernel that starts the transform reads the file's bytes directly. */
int khg_encode_bgr_dev(khg_ctx *ctx, const uint8_t *d_pixels, size_t stride, int rows, int cols, int lvl,
                       uint8_t *d_out, size_t cap, void *stream);
/* Waits for the context's last encode and returns its status and .comp length. */
int khg_encode_result(khg_ctx *ctx, size_t *out_len);

/* ---- decode: LoadComp's codec part (Main.cpp:283-315): MeasureImageSize (Reader.h:97-110),
 * ReadAll (Reader.h:56-74), DecompressImage (Decompress.h:7-126, 274-280, visualize=false),
 * YCbCr2RGB (Colorspace.h:25-36) and the double -> u8 truncation of Main.cpp:307-315. */
int khg_peek(const uint8_t *comp, size_t len, khg_header *hdr);  /* host bytes; validates the header */
int khg_decode(khg_ctx *ctx, const uint8_t *comp, size_t len, uint8_t *r, uint8_t *g, uint8_t *b);
/* rows/cols are what the caller expects the file to hold (checked against the header on the device).
 * d_comp must be 4-byte aligned, d_r / d_g / d_b 8-byte aligned (KHG_ERR_ARG otherwise). */
int khg_decode_dev(khg_ctx *ctx, const uint8_t *d_comp, size_t len, int rows, int cols,
                   uint8_t *d_r, uint8_t *d_g, uint8_t *d_b, void *stream);
/* khg_decode_dev followed by the pixel loop of CKhuGleSignal::SaveBmp (KhuGleSignal.cpp:285-293), fused into the last
 * inverse-transform kernel: the picture is written as the pixel array of a 24-bit BMP (B, G, R per pixel, image row y
 * at row rows - 1 - y of the array, `stride` >= 3 * cols bytes per row; bytes of a row beyond 3 * cols are left as they
 * are).  d_pixels 2-byte aligned, stride even.  With khugle_amd.bmp's header in front, the array is the file SaveBmp
 * writes. */
int khg_decode_bgr_dev(khg_ctx *ctx, const uint8_t *d_comp, size_t len, int rows, int cols, uint8_t *d_pixels,
                       size_t stride, void *stream);
/* Waits for the context's last enqueued work and returns its device-side status. */
int khg_last_status(khg_ctx *ctx);

/* ---- the reference's own entry points, which work on f64 Y / Cb / Cr planes (Y rows x cols, Cb and Cr
 * rows/2 x cols/2, contiguous): CompressImage (Compress.h:104-408) without RGB2YCbCr in front, and DecompressImage
 * (Decompress.h:7-126, 274-280) without YCbCr2RGB behind.  write_back != 0 leaves the DWT coefficients in the input
 * planes as CompressImage does (Compress.h:108-114).  The *_dev forms take device pointers (optional coefficient
 * planes, all three or none) and are asynchronous like khg_encode_dev / khg_decode_dev. */
int khg_encode_ycbcr(khg_ctx *ctx, double *y, double *cb, double *cr, int rows, int cols, int lvl, uint8_t *out,
                     size_t cap, size_t *out_len, int write_back);
int khg_decode_ycbcr(khg_ctx *ctx, const uint8_t *comp, size_t len, double *y, double *cb, double *cr);
int khg_encode_ycbcr_dev(khg_ctx *ctx, const double *d_y, const double *d_cb, const double *d_cr, int rows, int cols,
                         int lvl, uint8_t *d_out, size_t cap, double *d_coef_y, double *d_coef_cb, double *d_coef_cr,
                         void *stream);
int khg_decode_ycbcr_dev(khg_ctx *ctx, const uint8_t *d_comp, size_t len, int rows, int cols, double *d_y, double *d_cb,
                         double *d_cr, void *stream);

/* ---- quality metrics: ComputeSSIM (SSIM.h:57-72; the application passes the R channel only,
 * Main.cpp:250-253) and GetPsnr (KhuGleBase.cpp:613-627).  Device planes, synchronous result.
 * Per-block SSIM values are bit-exact; *ssim is their sum in raster order divided by the count (SSIM.h:66-71),
 * reproduced on the device with every rounding of that one-at-a-time sum (see khg_ordered_mean_dev), so it is
 * bit-identical too. */
int khg_ssim_dev(khg_ctx *ctx, const uint8_t *d_a, const uint8_t *d_b, int width, int height, double *ssim,
                 void *stream);
int khg_psnr_dev(khg_ctx *ctx, const uint8_t *d_ir, const uint8_t *d_ig, const uint8_t *d_ib,
                 const uint8_t *d_or, const uint8_t *d_og, const uint8_t *d_ob, int width, int height, double *psnr,
                 void *stream);
/* The last step of ComputeSSIM on its own (SSIM.h:66-71: the per-block values added one after another in raster
 * order, then divided by their number): d_mean[0] = (((v[0] + v[1]) + v[2]) + ...) / n with every rounding of that
 * loop, for any n >= 1 doubles in device memory.  Asynchronous on `stream`, no context, no allocation.  khg_ssim_dev
 * ends with this kernel. */
int khg_ordered_mean_dev(const double *d_values, size_t n, double *d_mean, void *stream);

/* ---- format 2: the reference's file followed by a seek index.  No counterpart in the reference (its format has no
 * synchronisation points; a decoder must first find where codewords begin).  After khg_ctx_set_format(ctx, 2) the
 * whole-image encoders of the context (khg_encode*, not the khg_tile_* steps) append, behind the unchanged bytes of
 * format 1 -- which ReadAll (Reader.h:56-74) reads as before, never looking further -- where the first codeword of
 * every 256 bits of each stream starts and which symbol it is, plus both dimensions and the stream lengths (layout:
 * csrc/khg_index.h; the length fields are 64 bits wide, but encoder and decoder keep the reference's limit of 2^31
 * bits per stream, Type.h:5, in both formats: a longer stream is KHG_ERR_RANGE, and an index whose high words are
 * not zero is refused).  The output buffer then needs khg_comp_bound + khg_index_bound bytes (KHG_ERR_CAP otherwise).
 * khg_decode_indexed_dev is khg_decode_dev for such files: it starts its decoders from the index instead of searching
 * for the codeword boundaries first (about a third of the decoder's work); a file without a sound index is refused
 * (KHG_ERR_FORMAT through khg_last_status).  An index that lies about a sound stream yields wrong pixels, never an
 * access outside the buffers.  khg_decode* ignore the index.  Format 1 is the default. */
int khg_ctx_set_format(khg_ctx *ctx, int format);
size_t khg_index_bound(int rows, int cols);
int khg_decode_indexed_dev(khg_ctx *ctx, const uint8_t *d_comp, size_t len, int rows, int cols, uint8_t *d_r,
                           uint8_t *d_g, uint8_t *d_b, void *stream);

/* ---- what the application shows beside a result (no part of the .comp file).
 * khg_status_strings: the two status lines of CompressImage / DecompressImage (Compress.h:294-300,
 * Decompress.h:265-271): "Step Size (min, max) : %.3f, %.3f" -- the extremes of the seven band steps at preset 0, of
 * the per-position divisor over the luma plane at presets 1 and 2 -- and "RLE Size (Y, Cb, Cr) : %d, %d, %d" (the
 * three table sizes, khg_header's rle_*_len).  Host only; either string may be NULL. */
int khg_status_strings(int rows, int cols, int lvl, int k_y, int k_cb, int k_cr, char *step_str, size_t step_cap,
                       char *rle_str, size_t rle_cap);
/* The three pictures, as planar u8 R, G, B in device memory (rows x cols each), asynchronous on `stream`:
 * - khg_vis_coeffs_dev: the DWT coefficient planes of the picture pushed through YCbCr2RGB (Compress.h:108-141);
 * - khg_vis_dequant_dev: the same of a .comp file's dequantised coefficients (Decompress.h:49-126, 234-262), read
 *   like khg_decode_dev (status through khg_last_status);
 * - khg_vis_steps_dev: the quantiser's step sizes in red (Compress.h:245-288, Decompress.h:190-232);
 * - khg_vis_bars_dev: the three stream lengths, in bytes, as bars (Compress.h:330-391 with the padded lengths,
 *   Decompress.h:128-188 with the pad bits taken off first).  Where the reference would draw outside the picture
 *   (rows > cols, or a stream of more than rows * rows bytes) nothing is drawn. */
int khg_vis_coeffs_dev(khg_ctx *ctx, const uint8_t *d_r, const uint8_t *d_g, const uint8_t *d_b, int rows, int cols,
                       uint8_t *d_out_r, uint8_t *d_out_g, uint8_t *d_out_b, void *stream);
int khg_vis_dequant_dev(khg_ctx *ctx, const uint8_t *d_comp, size_t len, int rows, int cols, uint8_t *d_out_r,
                        uint8_t *d_out_g, uint8_t *d_out_b, void *stream);
int khg_vis_steps_dev(int rows, int cols, int lvl, uint8_t *d_out_r, uint8_t *d_out_g, uint8_t *d_out_b, void *stream);
int khg_vis_bars_dev(int rows, int cols, const long long *len_bytes, uint8_t *d_out_r, uint8_t *d_out_g,
                     uint8_t *d_out_b, void *stream);
/* DownSampleImage (DownSample.h:5-37; the application draws every picture as a 256 x 256 thumbnail, Main.cpp:59-61):
 * one plane, bilinear interpolation in single precision, operation for operation.  Where the reference divides by zero
 * (sample positions on the source's last row or column) and then converts NaN to unsigned char, the value is 0, as in
 * the reference's x86-64 build.  Sides up to 2^22.  Asynchronous, no context. */
int khg_downsample_dev(const uint8_t *d_src, int src_rows, int src_cols, uint8_t *d_dst, int dst_rows, int dst_cols,
                       void *stream);

/* ---- an entry point that allocates nothing.  It has no counterpart in the reference: it exists for callers that
 * cannot allocate device or pinned memory -- code under HIP graph capture, tight serving loops, harnesses that
 * forbid hipMalloc -- and therefore cannot hold a khg_ctx either.  It is khg_ssim_dev with its scratch (the per-block
 * values) in a fixed array that is part of the library's code object on every GPU: at most 7680 x 4320 pixels
 * (518400 blocks, 4 MB per GPU until the process ends); larger images are refused with KHG_ERR_DIMS.  It works on
 * the calling thread's current device, enqueues on `stream` and does not synchronise: the result is written to
 * DEVICE memory at d_ssim[0] (bit-identical to khg_ssim_dev's) and is valid once the stream has run.  No hipMalloc,
 * hipFree or other allocating call is made.  All calls on one GPU share the one array: calls that overlap in time
 * on a GPU (different streams) race; nothing in it needs to start at zero.
 * The array's address on a device is looked up (hipGetSymbolAddress, which may load the code object) the first time
 * the device is used: that lookup must not happen while a stream is capturing.  Call khg_ssim_static_prepare() --
 * set-up, no compute: it does the lookup for the calling thread's current device -- or make the first
 * khg_ssim_static_dev call on the device, before capture begins.  `stream` must belong to the current device
 * (KHG_ERR_ARG otherwise). */
int khg_ssim_static_prepare(void);
int khg_ssim_static_dev(const uint8_t *d_a, const uint8_t *d_b, int width, int height, double *d_ssim, void *stream);

/* ---- parity taps (device pointers, synchronous).  Any output pointer may be NULL.
 * khg_forward_planes_dev: Y/Cb/Cr after the two FWT2D levels (Compress.h:108-114), f64, row-major.
 * khg_symbols_dev: the quantised planes (Compress.h:165-243) as int16, row-major. */
int khg_forward_planes_dev(khg_ctx *ctx, const uint8_t *d_r, const uint8_t *d_g, const uint8_t *d_b, int rows,
                           int cols, double *d_y, double *d_cb, double *d_cr, void *stream);
int khg_symbols_dev(khg_ctx *ctx, const uint8_t *d_r, const uint8_t *d_g, const uint8_t *d_b, int rows, int cols,
                    int lvl, int16_t *d_qy, int16_t *d_qcb, int16_t *d_qcr, void *stream);

/* ---- measurement aid: runs only the level-1 luma forward kernel (colour + DWT level 1 + quantiser +
 * histogram of the Y plane, the longest-running kernel of the path) `reps` times on `stream`,
 * bracketed by HIP events recorded on that stream, and returns the mean launch duration. */
int khg_time_luma_forward_dev(khg_ctx *ctx, const uint8_t *d_r, const uint8_t *d_g, const uint8_t *d_b, int rows,
                              int cols, int lvl, int reps, float *avg_ms, void *stream);

/* ---- batches of frames in host memory: LoadBMPAndCompress / LoadComp (Main.cpp:180-334) called in a loop, as a
 * pipeline.  `depth` contexts work on consecutive frames on streams of their own, so the PCIe copies of one frame
 * overlap the kernels of others; buffers the caller has pinned (hipHostMalloc / hipHostRegister) are used in
 * place, pageable ones go through pinned staging.  rgb[f] / rgb_out[f]: planar R, G, B of one frame; out[f] /
 * comp[f]: one .comp file image.  status[f] receives each frame's khg_status; the return value is the first
 * failure, KHG_OK if none. */
typedef struct khg_batch khg_batch;
int khg_batch_create(int device, int max_rows, int max_cols, int depth, khg_batch **out);
void khg_batch_destroy(khg_batch *b);
int khg_batch_encode(khg_batch *b, int n, const uint8_t *const *rgb, int rows, int cols, int lvl, uint8_t *const *out,
                     const size_t *cap, size_t *out_len, int *status);
int khg_batch_decode(khg_batch *b, int n, const uint8_t *const *comp, const size_t *len, uint8_t *const *rgb_out,
                     int *status);

/* ---- one image tiled by rows over several GPUs: one process and one context per GPU, rank t of n owns image
 * rows [rows*t/n, rows*(t+1)/n), rows a multiple of 8*n, n <= 8.  The reference has a single CompressImage call
 * (Compress.h:104) for this; the tiled form produces the same bytes.  The steps below do the per-GPU work; between
 * them the caller runs three collectives on buffers it owns (RCCL in this repository's host layer,
 * image-compression_amd/python/khugle_amd/tiled.py):
 *   1. khg_tile_forward_dev   colour + both DWT levels + quantiser on the tile (d_r/d_g/d_b hold image rows
 *                             row0 .. row1 of khg_tile_input_rows: the tile plus up to 24 halo rows per side)
 *   2. khg_tile_hist_get  ->  all-reduce SUM of the counts, MIN of the first raster indices (3 x 16384 words
 *                             each; 0xFFFFFFFF marks "absent")  ->  khg_tile_hist_set
 *   3. khg_tile_lens_dev      table, tree, codes (Huffman.h:87-139; identical on every rank), bits of own chunks
 *   4. khg_tile_bits_get  ->  all-reduce SUM over khg_tile_chunk_count words  ->  khg_tile_bits_set
 *   5. khg_tile_pack_dev      this rank's share of the .comp bytes at their final positions, zero elsewhere up to
 *                             the file's length (rank 0's share holds the header and tables; bytes behind the
 *                             file's length are left as they were); khg_encode_result: status, file length
 *   6. reduce SUM (or OR) of the shares as 32-bit words to the rank that keeps the file -- or, moving each byte
 *      once: khg_tile_chunk_off_get gives every chunk's bit offset in the file (the same on all ranks after step
 *      5), a rank's chunks are the ones whose bits were non-zero before step 4's all-reduce, so each rank sends the
 *      32-bit words its chunks touch and the keeper adds them in place (a word shared by two ranks' chunks gets
 *      both contributions: their bits are disjoint).
 * khg_tile_decode_dev: every rank decodes the whole stream and reconstructs its own rows (Decompress.h:7-126 on
 * the tile); d_r/d_g/d_b receive exactly the tile's rows. */
int khg_tile_input_rows(int rows, int rank, int nranks, int *row0, int *row1);
size_t khg_tile_chunk_count(int rows, int cols, int nranks);
int khg_tile_forward_dev(khg_ctx *ctx, const uint8_t *d_r, const uint8_t *d_g, const uint8_t *d_b, int rows, int cols,
                         int lvl, int rank, int nranks, void *stream);
int khg_tile_hist_get(khg_ctx *ctx, uint32_t *d_count, uint32_t *d_first, void *stream);
int khg_tile_hist_set(khg_ctx *ctx, const uint32_t *d_count, const uint32_t *d_first, void *stream);
int khg_tile_lens_dev(khg_ctx *ctx, void *stream);
int khg_tile_bits_get(khg_ctx *ctx, uint32_t *d_bits, void *stream);
int khg_tile_bits_set(khg_ctx *ctx, const uint32_t *d_bits, void *stream);
int khg_tile_pack_dev(khg_ctx *ctx, uint8_t *d_out, size_t cap, void *stream);
int khg_tile_chunk_off_get(khg_ctx *ctx, uint64_t *d_bit_offsets, void *stream);
int khg_tile_decode_dev(khg_ctx *ctx, const uint8_t *d_comp, size_t len, int rows, int cols, int rank, int nranks,
                        uint8_t *d_r, uint8_t *d_g, uint8_t *d_b, void *stream);

/* Tiled decode with the entropy decode (Huffman.h:33-55) split over the ranks instead of repeated on each: every
 * rank holds the whole file and decodes the spans [nb*t/n, nb*(t+1)/n) of each plane's stream.
 *   1. khg_tile_decode_map_dev    header, trees, and the start-offset maps of this rank's spans
 *   2. khg_tile_blk_get  ->  all-reduce SUM over khg_tile_blk_words 32-bit words (other ranks' records are zero)
 *                             ->  khg_tile_blk_set: every rank can now chain all spans from bit 0
 *   3. khg_tile_decode_write_dev  chains all spans, decodes this rank's spans into its symbol planes; returns (host
 *                             memory, after waiting for the stream) bounds[p*(n+1)+t] = raster index of the first
 *                             symbol of plane p that rank t holds, bounds[p*(n+1)+n] = symbols decoded in plane p
 *   4. point to point: each rank sends the parts of its symbols that lie in another rank's
 *                             khg_tile_need_ranges (raster ranges [first, last) of whole symbol rows, at most 4
 *                             per plane; returns their number) -- khg_tile_sym_get on the sender, khg_tile_sym_set
 *                             on the receiver
 *   5. khg_tile_decode_inverse_dev  dequantiser, inverse DWT, colour conversion of the tile's rows
 *                             (Decompress.h:56-126, 274-280); khg_last_status: the job's status. */
int khg_tile_decode_map_dev(khg_ctx *ctx, const uint8_t *d_comp, size_t len, int rows, int cols, int rank, int nranks,
                            void *stream);
size_t khg_tile_blk_words(khg_ctx *ctx);
int khg_tile_blk_get(khg_ctx *ctx, uint32_t *d_words, void *stream);
int khg_tile_blk_set(khg_ctx *ctx, const uint32_t *d_words, void *stream);
int khg_tile_decode_write_dev(khg_ctx *ctx, uint32_t *bounds, void *stream);
int khg_tile_need_ranges(int rows, int cols, int rank, int nranks, int plane, uint32_t *ranges);
int khg_tile_sym_get(khg_ctx *ctx, int plane, uint32_t first, uint32_t count, int16_t *d_dst, void *stream);
int khg_tile_sym_set(khg_ctx *ctx, int plane, uint32_t first, uint32_t count, const int16_t *d_src, void *stream);
int khg_tile_decode_inverse_dev(khg_ctx *ctx, uint8_t *d_r, uint8_t *d_g, uint8_t *d_b, void *stream);

/* ---- the tiled mode with its collectives inside the library (no counterpart in the reference, whose single
 * CompressImage / decompress calls, Compress.h:104 and Decompress.h:274, these replace for an image spread over
 * several GPUs).  khg_tile_encode and khg_tile_decode run the steps above and, between them, the exchanges -- the
 * all-reduces of the histogram, the chunk bit counts, the workgroup records and the status, the merge of the output
 * bytes, the point-to-point moves of halo rows and decoded symbols -- in place on the context's own buffers, on a
 * communicator:
 *   khg_comm_create_rccl   RCCL, one process per GPU: rank 0 calls khg_comm_unique_id (128 bytes), hands the id to
 *                          the other ranks by whatever means the application has, and every rank creates its
 *                          communicator from it (a collective call; `device` is made and left the calling thread's
 *                          current device).  khg_comm_wrap_rccl takes an ncclComm_t the
 *                          application already owns.  RCCL is looked up at run time (the copy the process holds,
 *                          else librccl.so.1): KHG_ERR_HIP if there is none.
 *   khg_comm_create_local  the ranks are threads of one process with their contexts on one GPU
 *                          (khg_comm_local_group(n) makes the group, each thread joins as one rank): the same host
 *                          code with device-to-device copies in place of the links -- for rehearsals and tests.
 * At most 8 ranks.  All ranks call each function together, with the same rows / cols / lvl / root / merge, and
 * contexts of the same capacity.  Both functions wait for the stream.  A failure that one rank's checks or kernels
 * find (bad argument, busy context, symbol out of range, output too small, damaged stream) is returned by every
 * rank -- the one that found it returns its own code, the others the lowest code any rank had; KHG_ERR_HIP (also
 * for RCCL errors) is returned at once and leaves the communicator unusable.
 *   khg_tile_exchange_halos  d_own_*: the rank's rows/n image rows (rows/n >= 24); d_strip_*: receives image rows
 *                          khg_tile_input_rows -- the rank's rows with its neighbours' 24 edge rows attached.
 *   khg_tile_encode        d_r/d_g/d_b: that strip.  The file -- the bytes one encoder produces -- arrives in d_out
 *                          on rank `root` (on other ranks d_out is scratch of the same capacity, khg_comp_bound);
 *                          *out_len on every rank.  merge 0: one reduce of the file's size adds the ranks' zero-padded
 *                          shares; merge 1: every byte moves once, point to point (a kernel on each rank works out
 *                          which 32-bit words its chunks touch and packs them; the keeper adds them in place).
 *   khg_tile_decode        d_comp: the whole file on every rank; d_r/d_g/d_b receive the rank's rows.  The entropy
 *                          decode is split over the ranks; decoded symbols go straight from the symbol planes of the
 *                          ranks that hold them into those of the ranks whose rows need them. */
typedef struct khg_comm khg_comm;
int khg_comm_unique_id(void *id, size_t bytes);
int khg_comm_create_rccl(khg_comm **out, int rank, int nranks, const void *id, size_t bytes, int device);
int khg_comm_wrap_rccl(khg_comm **out, void *nccl_comm, int rank, int nranks);
void *khg_comm_local_group(int nranks);
int khg_comm_create_local(khg_comm **out, void *group, int rank);
void khg_comm_destroy(khg_comm *comm);
int khg_tile_exchange_halos(khg_comm *comm, const uint8_t *d_own_r, const uint8_t *d_own_g, const uint8_t *d_own_b,
                            int rows, int cols, uint8_t *d_strip_r, uint8_t *d_strip_g, uint8_t *d_strip_b,
                            void *stream);
int khg_tile_encode(khg_ctx *ctx, khg_comm *comm, const uint8_t *d_r, const uint8_t *d_g, const uint8_t *d_b, int rows,
                    int cols, int lvl, uint8_t *d_out, size_t cap, size_t *out_len, int root, int merge, void *stream);
int khg_tile_decode(khg_ctx *ctx, khg_comm *comm, const uint8_t *d_comp, size_t len, int rows, int cols, uint8_t *d_r,
                    uint8_t *d_g, uint8_t *d_b, void *stream);

/* Measurement aid: while enabled, khg_encode_dev brackets its luma level-1 forward launch with HIP events on
 * the caller's stream, and the kernel notes the device clock at which its first workgroup starts and its last
 * one ends; khg_encode_result adds both durations to per-context tallies that khg_profile_read returns
 * (event-to-event milliseconds, which include the wait for the GPU when other streams are busy; device-clock
 * milliseconds, which are the launch's execution span as a profiler reports it; number of launches).
 * Enabling or disabling resets the tallies. */
int khg_profile_luma_forward(khg_ctx *ctx, int enable);
int khg_profile_read(khg_ctx *ctx, double *total_ms, double *total_device_ms, int *launches);

/* ---- test hook (no counterpart in the reference): fills the context's whole device workspace with `byte` and
 * waits for the fill.  Parity tests call it between jobs to show that no job reads what an earlier job (of the
 * same or another image size) left behind.  KHG_ERR_BUSY while a result is unfetched. */
int khg_debug_poison(khg_ctx *ctx, int byte);

/* ---- synthetic frames for benchmarks: deterministic integer arithmetic, identical to
 * khugle_amd.synth.synth_frame on the host.  Device planes. */
int khg_synth_dev(uint8_t *d_r, uint8_t *d_g, uint8_t *d_b, int rows, int cols, uint32_t seed, uint32_t index,
                  void *stream);
/* Rows row0 .. row0+rows-1 of the same frame (a tile of a large image is generated where it is used). */
int khg_synth_rows_dev(uint8_t *d_r, uint8_t *d_g, uint8_t *d_b, int row0, int rows, int cols, uint32_t seed,
                       uint32_t index, void *stream);

#ifdef __cplusplus
}
#endif
#endif
