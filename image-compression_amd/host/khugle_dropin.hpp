// khugle_dropin.hpp -- the reference's codec entry points, served by libkhugle_hip.so (MI355X).
//
// Host-side mirror of the interface the KhuGle application calls (Main.cpp:196-208, 283-315):
//     CompressImage   (Compress.h:104)    f64 Y / Cb / Cr planes as double**  ->  CompResult
//     DecompressImage (Decompress.h:7)    CompResult  ->  f64 Y / Cb / Cr planes
//     WriteAll / ReadAll (Writer.h:43, Reader.h:56)      CompResult  <->  .comp file
// Same names, same argument meaning, same data structures (CompResult keeps its '0'/'1' bit strings including
// the pad bits and its (count, symbol) tables, so the reference's own WriteAll / ReadAll and this header's are
// interchangeable).  Differences, all deliberate:
//   * the GUI by-products of CompressImage / DecompressImage (visualisation images, status strings) are not
//     produced; the overloads that take those arguments leave them untouched;
//   * where the reference fails silently or reads out of bounds, these functions throw std::runtime_error
//     carrying the library's status text (there is no CPU path behind them);
//   * rows and columns may differ (header offset 24 = rows, 28 = cols; identical for the reference's squares).
//
// Inside the reference tree: #include "Type.h" first and define KHUGLE_DROPIN_USE_REFERENCE_TYPES; the functions
// then work on the application's own HeaderInfo / CompResult.  Stand-alone (tests, other callers) the header
// declares layout-compatible types of the same names in namespace khugle_hip.
#pragma once

#include <cstdint>
#include <cstring>
#include <fstream>
#include <iterator>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "khugle_hip.h"

namespace khugle_hip {

#ifdef KHUGLE_DROPIN_USE_REFERENCE_TYPES
using ::CompResult;
using ::HeaderInfo;
#else
// Type.h:3-28 / 30-51: member for member.
struct HeaderInfo {
    int y_len = 0, cb_len = 0, cr_len = 0;              // padded stream lengths in bits
    int rle_y_len = 0, rle_cb_len = 0, rle_cr_len = 0;  // table sizes
    int h = 0, w = 0;
    unsigned char y_pad = 0, cb_pad = 0, cr_pad = 0;
    unsigned char lvl = 0;
};

class CompResult {
public:
    HeaderInfo info_;
    std::string encoded_data_y_, encoded_data_cb_, encoded_data_cr_;
    std::vector<std::pair<int, int>> rle_y_, rle_cb_, rle_cr_;
    CompResult(HeaderInfo info, std::string y, std::string cb, std::string cr, std::vector<std::pair<int, int>> ry,
               std::vector<std::pair<int, int>> rcb, std::vector<std::pair<int, int>> rcr)
        : info_(info), encoded_data_y_(std::move(y)), encoded_data_cb_(std::move(cb)), encoded_data_cr_(std::move(cr)),
          rle_y_(std::move(ry)), rle_cb_(std::move(rcb)), rle_cr_(std::move(rcr)) {}
};
#endif

namespace detail {

inline void check(int status, const char* what)
{
    if (status != KHG_OK) throw std::runtime_error(std::string(what) + ": " + khg_status_string(status));
}

// One library context per thread, grown when a larger image comes along.
inline khg_ctx* session(int rows, int cols)
{
    struct Holder {
        khg_ctx* ctx = nullptr;
        int rows = 0, cols = 0;
        ~Holder() { if (ctx) khg_ctx_destroy(ctx); }
    };
    static thread_local Holder h;
    if (!h.ctx || rows > h.rows || cols > h.cols) {
        if (h.ctx) khg_ctx_destroy(h.ctx);
        h.ctx = nullptr;
        h.rows = rows > h.rows ? rows : h.rows;
        h.cols = cols > h.cols ? cols : h.cols;
        check(khg_ctx_create(0, h.rows, h.cols, &h.ctx), "khg_ctx_create");
    }
    return h.ctx;
}

inline int32_t le32(const uint8_t* p) { int32_t v; std::memcpy(&v, p, 4); return v; }  // gfx950 hosts are little-endian
inline void put32(std::vector<uint8_t>& o, int32_t v) { const uint8_t* p = reinterpret_cast<const uint8_t*>(&v); o.insert(o.end(), p, p + 4); }

inline std::string bits_of(const uint8_t* p, size_t nbytes)
{
    std::string s(nbytes * 8, '0');
    for (size_t i = 0; i < nbytes; ++i)
        for (int b = 0; b < 8; ++b)
            if (p[i] & (0x80u >> b)) s[i * 8 + b] = '1';
    return s;
}

inline void bytes_of(const std::string& bits, std::vector<uint8_t>& out)
{
    for (size_t i = 0; i + 8 <= bits.size(); i += 8) {
        unsigned v = 0;
        for (int b = 0; b < 8; ++b) v = (v << 1) | (bits[i + b] == '1');
        out.push_back(static_cast<uint8_t>(v));
    }
}

inline void flatten(double** rows, int h, int w, std::vector<double>& out)
{
    out.resize(static_cast<size_t>(h) * w);
    for (int y = 0; y < h; ++y) std::memcpy(&out[static_cast<size_t>(y) * w], rows[y], sizeof(double) * w);
}

inline void unflatten(const std::vector<double>& in, double** rows, int h, int w)
{
    for (int y = 0; y < h; ++y) std::memcpy(rows[y], &in[static_cast<size_t>(y) * w], sizeof(double) * w);
}

}  // namespace detail

// The bytes WriteAll puts on disk (Writer.h:9-69).
inline std::vector<uint8_t> Serialize(const CompResult& r)
{
    std::vector<uint8_t> o;
    const HeaderInfo& i = r.info_;
    for (int v : {i.y_len, i.cb_len, i.cr_len, i.rle_y_len, i.rle_cb_len, i.rle_cr_len, i.h, i.w}) detail::put32(o, v);
    o.push_back(i.y_pad); o.push_back(i.cb_pad); o.push_back(i.cr_pad); o.push_back(i.lvl);
    detail::bytes_of(r.encoded_data_y_, o);
    detail::bytes_of(r.encoded_data_cb_, o);
    detail::bytes_of(r.encoded_data_cr_, o);
    for (const auto* t : {&r.rle_y_, &r.rle_cb_, &r.rle_cr_})
        for (const auto& e : *t) { detail::put32(o, e.first); detail::put32(o, e.second); }
    return o;
}

// What ReadAll builds from those bytes (Reader.h:5-74), after the library has validated them.
inline std::unique_ptr<CompResult> Parse(const uint8_t* p, size_t len)
{
    khg_header hd;
    detail::check(khg_peek(p, len, &hd), "khg_peek");
    HeaderInfo i;
    i.y_len = hd.y_len; i.cb_len = hd.cb_len; i.cr_len = hd.cr_len;
    i.rle_y_len = hd.rle_y_len; i.rle_cb_len = hd.rle_cb_len; i.rle_cr_len = hd.rle_cr_len;
    i.h = hd.h; i.w = hd.w;
    i.y_pad = hd.y_pad; i.cb_pad = hd.cb_pad; i.cr_pad = hd.cr_pad; i.lvl = hd.lvl;
    size_t off = 36;
    std::string s[3];
    const int lens[3] = {i.y_len, i.cb_len, i.cr_len};
    for (int k = 0; k < 3; ++k) { s[k] = detail::bits_of(p + off, static_cast<size_t>(lens[k]) / 8); off += static_cast<size_t>(lens[k]) / 8; }
    std::vector<std::pair<int, int>> t[3];
    const int ks[3] = {i.rle_y_len, i.rle_cb_len, i.rle_cr_len};
    for (int k = 0; k < 3; ++k)
        for (int e = 0; e < ks[k]; ++e, off += 8) t[k].emplace_back(detail::le32(p + off), detail::le32(p + off + 4));
    return std::make_unique<CompResult>(i, s[0], s[1], s[2], t[0], t[1], t[2]);
}

// Compress.h:104.  InputY is img_height x img_width, InputCb / InputCr are half that in both directions; like the
// reference, the call leaves the DWT coefficients in the three input planes.
inline void CompressImage(std::unique_ptr<CompResult>& res, double** InputY, double** InputCb, double** InputCr,
                          int img_height, int img_width, int lvl)
{
    khg_ctx* ctx = detail::session(img_height, img_width);
    std::vector<double> y, cb, cr;
    detail::flatten(InputY, img_height, img_width, y);
    detail::flatten(InputCb, img_height / 2, img_width / 2, cb);
    detail::flatten(InputCr, img_height / 2, img_width / 2, cr);
    std::vector<uint8_t> out(khg_comp_bound(img_height, img_width));
    size_t len = 0;
    detail::check(khg_encode_ycbcr(ctx, y.data(), cb.data(), cr.data(), img_height, img_width, lvl, out.data(),
                                   out.size(), &len, 1), "khg_encode_ycbcr");
    detail::unflatten(y, InputY, img_height, img_width);
    detail::unflatten(cb, InputCb, img_height / 2, img_width / 2);
    detail::unflatten(cr, InputCr, img_height / 2, img_width / 2);
    res = Parse(out.data(), len);
}

namespace detail {
// The two status lines of Compress.h:294-300 / Decompress.h:265-271.
inline void status_lines(const HeaderInfo& i, std::string& str, std::string& str2)
{
    char a[100], b[100];
    check(khg_status_strings(i.h, i.w, i.lvl, i.rle_y_len, i.rle_cb_len, i.rle_cr_len, a, sizeof a, b, sizeof b),
          "khg_status_strings");
    str = a;
    str2 = b;
}
}  // namespace detail

// The application's full argument list (Compress.h:104-106): str and str2 receive the two status lines;
// out_image_vec is left alone (its pictures come from khg_vis_coeffs_dev / khg_vis_steps_dev / khg_vis_bars_dev as
// planar u8 device memory, which the caller copies into its CKhuGleSignal rows).
template <class ImageVec>
inline void CompressImage(std::unique_ptr<CompResult>& res, double** InputY, double** InputCb, double** InputCr,
                          int img_height, int img_width, int lvl, ImageVec&, std::string& str, std::string& str2)
{
    CompressImage(res, InputY, InputCb, InputCr, img_height, img_width, lvl);
    detail::status_lines(res->info_, str, str2);
}

// Decompress.h:7 with visualize = false.
inline void DecompressImage(const CompResult* result, double** OutY, double** OutCb, double** OutCr)
{
    if (!result) throw std::runtime_error("DecompressImage: null CompResult");
    const std::vector<uint8_t> bytes = Serialize(*result);
    const int h = result->info_.h, w = result->info_.w;
    khg_ctx* ctx = detail::session(h, w);
    std::vector<double> y(static_cast<size_t>(h) * w), cb(y.size() / 4), cr(y.size() / 4);
    detail::check(khg_decode_ycbcr(ctx, bytes.data(), bytes.size(), y.data(), cb.data(), cr.data()), "khg_decode_ycbcr");
    detail::unflatten(y, OutY, h, w);
    detail::unflatten(cb, OutCb, h / 2, w / 2);
    detail::unflatten(cr, OutCr, h / 2, w / 2);
}

// Writer.h:43 / Reader.h:56.
inline void WriteAll(const char* write_path, CompResult* result)
{
    std::ofstream f(write_path, std::ios::out | std::ios::binary);
    if (!f) throw std::runtime_error(std::string("WriteAll: cannot open ") + write_path);
    const std::vector<uint8_t> b = Serialize(*result);
    f.write(reinterpret_cast<const char*>(b.data()), static_cast<std::streamsize>(b.size()));
}

inline CompResult* ReadAll(const char* read_path)
{
    std::ifstream f(read_path, std::ios::in | std::ios::binary);
    if (!f) throw std::runtime_error(std::string("ReadAll: cannot open ") + read_path);
    const std::vector<uint8_t> b((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    return Parse(b.data(), b.size()).release();
}

// Decompress.h:285 with the GUI arguments dropped.
inline void DecompressImage(const char* read_path, double** OutY, double** OutCb, double** OutCr)
{
    std::unique_ptr<CompResult> r(ReadAll(read_path));
    DecompressImage(r.get(), OutY, OutCb, OutCr);
}

// Decompress.h:7 / 285 with the GUI arguments: the status lines as above when visualize is set.
template <class ImageVec>
inline void DecompressImage(const CompResult* result, double** OutY, double** OutCb, double** OutCr, bool visualize,
                            ImageVec&, std::string& str, std::string& str2)
{
    DecompressImage(result, OutY, OutCb, OutCr);
    if (visualize) detail::status_lines(result->info_, str, str2);
}
template <class ImageVec>
inline void DecompressImage(const char* read_path, double** OutY, double** OutCb, double** OutCr, ImageVec& v,
                            std::string& str, std::string& str2)
{
    std::unique_ptr<CompResult> r(ReadAll(read_path));
    DecompressImage(r.get(), OutY, OutCb, OutCr, true, v, str, str2);
}

}  // namespace khugle_hip
